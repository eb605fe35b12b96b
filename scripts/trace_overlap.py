import csv,sys,glob
fn=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(fn)) if 'ga::' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
for r in rows[-12:]:
    print(r['Kernel_Name'].split('(')[0][-40:], (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-t0)/1e3, r.get('Stream_Id',''))
