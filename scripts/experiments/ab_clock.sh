#!/bin/bash
# GPU box: shader clock during k_isab1_fwd256_ab (GRBM_GUI_ACTIVE cycles / kernel duration) for the
# shipped kernel and for ablated measurement builds - is the kernel power-limited (clock rises when
# work is removed)?
set -e
R=$GRAFT_REPO_ROOT
cd $R
SO=point-cloud-audio_amd/pca_hip/libpca_hip.so
cp $SO /tmp/lib_p.so
touch point-cloud-audio_amd/csrc/d256_fused.hip
HIPCC="/opt/rocm/bin/hipcc -DPCA_FWD_ABLATE" bash point-cloud-audio_amd/csrc/build.sh > /dev/null
cp $SO /tmp/lib_a.so
cd /tmp && export TMPDIR=/tmp
run() {   # lib mask
  cp /tmp/lib_$1.so $R/$SO
  rm -rf /tmp/clk
  PCA_AB_ABLATE=$2 NS=2048 REPS=20 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/clk -- python3 $R/scripts/fwd256_bench.py > /tmp/clk.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("/tmp/clk/**/*counter_collection.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "k_isab1_fwd256_ab" in r["Kernel_Name"]]
tr=glob.glob("/tmp/clk/**/*kernel_trace.csv",recursive=True)[0]
d={r["Dispatch_Id"]:(int(r["End_Timestamp"])-int(r["Start_Timestamp"])) for r in csv.DictReader(open(tr))}
cyc=[float(r["Counter_Value"]) for r in rows]; ns=[d[r["Dispatch_Id"]] for r in rows]
n=len(cyc)//2
c=sum(cyc[n:])/len(cyc[n:]); t=sum(ns[n:])/len(ns[n:])
print("lib $1 ablate=$2: %.1f us, %.0f cycles -> %.2f GHz"%(t/1e3,c,c/t))
PY
}
run p 0
run a 0
run a 16
run a 96
run a 127
run p 0
cp /tmp/lib_p.so $R/$SO
