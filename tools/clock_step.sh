#!/bin/bash
# Effective shader clock and matrix-pipe occupancy of every kernel of one eager training step:
#   rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace over tools/pmc_step.py
# clock ~ GRBM_GUI_ACTIVE / 8 XCDs / duration (MI355X_MICROARCH.md, DVFS give-back; reads high on dispatches < 0.3 ms)
# usage: bash tools/clock_step.sh [config] [dtype] [trees]   -> gpurun_out/clock/summary.txt
C=${1:-st_pgat_spgnn_3}; DT=${2:-f32}; T=${3:-512}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/clock; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(cd $R && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/p -- python3 tools/pmc_step.py $C $DT $T $O/manifest.json > $O/p.log 2>&1) || { echo "pass failed"; tail -5 $O/p.log; exit 1; }
cd $R && python3 - <<P
import glob, pandas as pd
O="$O"
c=pd.concat([pd.read_csv(f) for f in glob.glob(O+'/p/**/*counter_collection.csv', recursive=True)])
k=pd.concat([pd.read_csv(f) for f in glob.glob(O+'/p/**/*kernel_trace.csv', recursive=True)])
k['dur_us']=(k.End_Timestamp-k.Start_Timestamp)/1e3
w=c.pivot_table(index=['Dispatch_Id','Kernel_Name'],columns='Counter_Name',values='Counter_Value',aggfunc='sum').reset_index()
m=w.merge(k[['Dispatch_Id','dur_us']],on='Dispatch_Id')
last=m.Dispatch_Id.max()
m=m[m.Dispatch_Id>last-140]                       # the last (instrumented) step
m['clk_GHz']=m.GRBM_GUI_ACTIVE/8/(m.dur_us*1e3)
m['name']=m.Kernel_Name.str.slice(0,48)
m['mfma_busy_per_cycle']=m.SQ_VALU_MFMA_BUSY_CYCLES/(m.GRBM_GUI_ACTIVE/8)
g=m.groupby('name').agg(n=('dur_us','size'),dur_us=('dur_us','mean'),clk_GHz=('clk_GHz','mean'),insts_mfma=('SQ_INSTS_MFMA','mean'),mfma_busy_per_cycle=('mfma_busy_per_cycle','mean')).sort_values('dur_us',ascending=False)
open(O+'/summary.txt','w').write(g.to_string())
print(g.head(30).to_string())
P
