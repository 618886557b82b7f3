import csv, sys, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[1:]:
    for row in csv.DictReader(open(f)):
        if 'sattn' in row['Kernel_Name']:
            acc[row['Kernel_Name'].split('<')[0].split('(')[0][:40]][row['Counter_Name']].append(float(row['Counter_Value']))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print('   %-28s %14.0f  (n=%d)' % (c, sum(v) / len(v), len(v)))
