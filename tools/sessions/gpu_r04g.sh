#!/bin/bash
# round-4 session 7: A/B of builds of the one-pass Nuth-Kaab kernel inside one session (kernel time from rocprofv3 --kernel-trace)
O=gpurun_out/r04j; mkdir -p $O
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
for lib in libxdemhip.so libxdemhip_nkw1.so libxdemhip_nkw2.so libxdemhip_nkw3.so libxdemhip_nkw4.so libxdemhip_nkw5.so libxdemhip_nkw6.so libxdemhip.so; do
  t=${lib%.so}; t=${t#libxdemhip}; t=${t:-_default}
  NK_LIB=$GRAFT_REPO_ROOT/xdem_amd/csrc/$lib timeout 150 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr$t$RANDOM -o nk -- python $GRAFT_REPO_ROOT/tools/nk_trace.py 20000 3 > $GRAFT_REPO_ROOT/$O/log$t.txt 2>&1
  grep "step 2" $GRAFT_REPO_ROOT/$O/log$t.txt | tail -2
done
cd $GRAFT_REPO_ROOT
python - <<'P'
import csv, glob, os
for d in sorted(glob.glob("gpurun_out/r04j/tr*"), key=os.path.getmtime):
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        ds = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if "nk_fused_kernel" in r["Kernel_Name"])
        print(os.path.basename(d), "nk_fused_kernel us:", [round(x) for x in ds])
P
find $O -name '*.csv' -size +1M -delete
