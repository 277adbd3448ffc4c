#!/bin/bash
# timing-only ablation of k_fast_blur_band phases (results are invalid under ARIA_ABLATE != 0)
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for a in 0 2 4 8 6 14; do
  rm -rf $R/gpurun_out/abl_$a
  ARIA_ABLATE=$a timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/abl_$a -- python3 $R/tools/prof_extract.py --pairs 512 --iters 4 --chunk 1024 > $R/gpurun_out/abl_$a.log 2>&1
done
python3 - <<'PY'
import pandas as pd, glob, os
R=os.environ["GRAFT_REPO_ROOT"]
rows={}
for a in [0,2,4,8,6,14]:
    f=sorted(glob.glob(f"{R}/gpurun_out/abl_{a}/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime)
    if not f: continue
    df=pd.read_csv(f[-1]); df['dur']=(df.End_Timestamp-df.Start_Timestamp)/1e3
    d=df[df.Kernel_Name.str.contains('fast_blur_band')]
    rows[a]=d.groupby(['Grid_Size_X','Workgroup_Size_X']).dur.median()
t=pd.DataFrame(rows).round(1); t.loc['sum']=t.sum(); t.loc['us/frame']=t.loc['sum']/1024; print(t.to_string())
PY
