"""Per-grid (= per pyramid level) averages of the kernels in a rocprofv3 kernel trace: python tools/ktrace_levels.py <kernel_trace.csv>"""
import csv,collections,sys
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    for key in sys.argv[2:]:
        if key in n:
            agg[(key,r['Grid_Size_X'],r['Grid_Size_Y'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(agg.items()):
    v=sorted(v,reverse=True)
    print(k,len(v),"median of top half",round(v[len(v)//4],1),[round(x,1) for x in v][:6])
