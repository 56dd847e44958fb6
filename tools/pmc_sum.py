import csv,glob,collections,sys
base='/root/repo/gpurun_out/prof_%s/'%sys.argv[1]
for f in glob.glob(base+'kt/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'gik' in r['Name']: print(r['Name'][:60], r['Calls'], r['AverageNs'])
tot=collections.defaultdict(lambda: collections.defaultdict(float))
for d in glob.glob(base+'pmc_*/'):
    for f in glob.glob(d+'*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            tot[r['Kernel_Name'][:40]][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in tot.items():
    if 'prep' in k: print(k, {a: round(b/3/1e6,1) for a,b in sorted(v.items()) if a in ('SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_INSTS_LDS','SQ_LDS_IDX_ACTIVE','SQ_LDS_BANK_CONFLICT','SQ_WAVE_CYCLES','SQ_ACTIVE_INST_LDS','SQ_WAIT_INST_ANY','SQ_BUSY_CYCLES')})
