import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'fa' in r['Kernel_Name']:
            print(r['Kernel_Name'].split('(')[1][:30] if 'anonymous' in r['Kernel_Name'] else r['Kernel_Name'][:40], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
