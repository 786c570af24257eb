import csv,glob,sys,collections
acc=collections.defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(d+"/**/*_counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            if "filter_rows_kernel" in r["Kernel_Name"] and "true, true, false" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
c={k:sum(v)/len(v) for k,v in acc.items()}
for k,v in sorted(c.items()): print(k, v, len(acc[k]))
if "SQ_WAVE_CYCLES" in c:
    for k in ("SQ_ACTIVE_INST_ANY","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_WAIT_INST_LDS","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS"):
        if k in c: print("share of wave cycles", k, round(c[k]/c["SQ_WAVE_CYCLES"],3))
if "SQ_LDS_IDX_ACTIVE" in c and "SQ_LDS_BANK_CONFLICT" in c: print("bank conflict share of LDS cycles", c["SQ_LDS_BANK_CONFLICT"]/c["SQ_LDS_IDX_ACTIVE"])
if "SQ_INSTS_VALU" in c: print("VALU wave-instr per thread-wave:", c["SQ_INSTS_VALU"]/(1024*4))
