#!/bin/bash
# PMC passes over tools/gemm_driver.py (NT / TN GEMMs at three shapes); one counter group per pass.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_gemm
mkdir -p $O
rocprofv3 -L > $O/avail.txt 2>&1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/p$i -- python3 $R/tools/gemm_driver.py > $O/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<'P'
import glob, pandas as pd, os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc_gemm'
rows=[]
for f in sorted(glob.glob(O+'/p*/**/*counter_collection.csv', recursive=True)):
    df=pd.read_csv(f)
    df=df[df.Kernel_Name.str.contains('gemm_nt_f16x3|gemm_tn_f16x3')]
    g=df.groupby(['Kernel_Name','Grid_Size','Counter_Name'],as_index=False)['Counter_Value'].mean()
    rows.append(g)
if rows:
    a=pd.concat(rows); a['Kernel_Name']=a.Kernel_Name.str.slice(0,40)
    a.to_csv(O+'/summary.csv',index=False); print(a.to_string())
P
