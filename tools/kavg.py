import csv,sys,glob
for f in glob.glob(sys.argv[1]+'/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'axvs' in r['Name'] and int(r['Calls'])>100: print('  ',r['Name'][11:50], r['AverageNs'][:8])
