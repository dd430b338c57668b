import csv,sys,glob
f=glob.glob('/tmp/pt/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last step: find the last 3 k_front_end_fused launches
fe=[i for i,r in enumerate(rows) if 'k_front_end_fused' in r['Kernel_Name']]
i0=fe[-3]
t0=int(rows[i0]['Start_Timestamp'])
for r in rows[i0:]:
    n=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:34]
    s=(int(r['Start_Timestamp'])-t0)/1e6; e=(int(r['End_Timestamp'])-t0)/1e6
    if e-s>0.08: print("%-36s q%-3s %7.3f -> %7.3f  (%6.3f)"%(n,r.get('Queue_Id','?'),s,e,e-s))
