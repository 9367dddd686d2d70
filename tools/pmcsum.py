
"""Sum rocprofv3 --pmc counters of the longest dispatch of a kernel.  usage: pmcsum.py <kernel substring> <dir>..."""
import csv, collections, sys, glob
pat = sys.argv[1]
for d in sys.argv[2:]:
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    dur = {}
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            agg[r['Dispatch_Id']][r['Counter_Name']] += float(r['Counter_Value'])
            dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    k = max(dur, key=dur.get)
    print(d, 'ms=%.3f' % dur[k], {a: round(b / 1e9, 4) for a, b in sorted(agg[k].items())})
