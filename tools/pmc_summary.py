"""Summarise rocprofv3 --pmc counter_collection.csv files: mean per dispatch of each counter for kernels matching a name."""
import csv, glob, sys, collections
pat = sys.argv[1]
for d in sys.argv[2:]:
    for f in glob.glob(d + '/*/*_counter_collection.csv'):
        acc = collections.defaultdict(list)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if pat in row['Kernel_Name']:
                    acc[row['Counter_Name']].append(float(row['Counter_Value']))
        for k, v in sorted(acc.items()):
            print('%-28s n=%3d mean=%.4g' % (k, len(v), sum(v) / len(v)))
