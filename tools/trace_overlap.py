"""How much kernel time of a rocprofv3 --kernel-trace run overlaps in time (parallel HIP-graph branches / streams).
usage: trace_overlap.py <rocprof dir>"""
import pandas as pd, glob, sys
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
df=pd.read_csv(f).sort_values('Start_Timestamp')
df=df.tail(3000)
ov=0; tot=0
prev_end=0
for s,e in zip(df.Start_Timestamp, df.End_Timestamp):
    if s<prev_end: ov+=min(e,prev_end)-s
    tot+=e-s
    prev_end=max(prev_end,e)
print('kernel time', tot/1e6,'ms; overlapped', ov/1e6,'ms; queues', df.Queue_Id.nunique(), df.Stream_Id.nunique() if 'Stream_Id' in df else '')
print(df.Queue_Id.value_counts().head())
