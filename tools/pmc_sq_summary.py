"""Per-kernel means of the SQ counters collected by tools/gpu_pmc_sq.sh -> csv on stdout (committed under profiles/)."""
import collections
import csv
import re
import sys

NAMES = ['SQ_WAVES', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_WAVE_CYCLES',
         'SQ_BUSY_CYCLES', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_ANY',
         'SQ_WAIT_INST_ANY', 'SQ_LDS_BANK_CONFLICT']


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'\(.*', '', n)


agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        k = short(r['Kernel_Name'])
        if k.startswith('at::') or k.startswith('__amd'):
            continue
        a = agg[k][r['Counter_Name']]
        a[0] += 1
        a[1] += float(r['Counter_Value'])
print('kernel,' + ','.join(NAMES))
key = lambda k: -agg[k].get('SQ_WAVE_CYCLES', [1, 0])[1] / max(1, agg[k].get('SQ_WAVE_CYCLES', [1, 0])[0])
for k in sorted(agg, key=key):
    print('"' + k + '",' + ','.join(f"{agg[k][n][1] / agg[k][n][0]:.0f}" if n in agg[k] else '' for n in NAMES))
