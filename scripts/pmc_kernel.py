"""Summarise a rocprofv3 --pmc csv: mean counter value per kernel name (substring filter)."""
import csv
import glob
import sys
from collections import defaultdict

root, needle = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: [0.0, 0])
for path in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            if needle not in row['Kernel_Name']:
                continue
            key = (row['Kernel_Name'][:70], row['Counter_Name'])
            acc[key][0] += float(row['Counter_Value'])
            acc[key][1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print(f'{k:70s} {c:28s} n={n:4d} mean={v / n:16.1f}')
