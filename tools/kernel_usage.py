"""Register / scratch / occupancy table of one translation unit's kernels, from hipcc's own remarks.

    hipcc <the Makefile's flags> -Rpass-analysis=kernel-resource-usage -c csrc/stripe_b4.hip -o /tmp/x.o 2> usage.txt
    python tools/kernel_usage.py usage.txt [substring of the mangled kernel name] [--spills]

Prints one line per kernel: the template arguments as they appear in the mangled name (Li<n>E / Lb<0|1>E), VGPRs, AGPRs, scratch
bytes per lane, VGPR spills, occupancy, LDS.  `--spills` keeps only kernels with scratch or spills: the check to run after adding
instantiations (a spilled instance of a latency-bound decode kernel costs a vmcnt(0) per reload)."""
import re
import sys


def parse(path):
    text = open(path, errors='replace').read()
    out = []
    for blk in re.split(r'remark: Function Name: ', text)[1:]:
        name = blk.split()[0]
        g = lambda key: int(re.search(re.escape(key) + r': (\d+)', blk).group(1))
        out.append(dict(name=name, args=re.findall(r'L([ib])(\d+)E', name), vgprs=g('VGPRs'), agprs=g('AGPRs'), scratch=g('ScratchSize [bytes/lane]'),
                        spill=g('VGPRs Spill'), occ=g('Occupancy [waves/SIMD]'), lds=g('LDS Size [bytes/block]')))
    return out


if __name__ == '__main__':
    rows = parse(sys.argv[1])
    sub = [a for a in sys.argv[2:] if not a.startswith('--')]
    only_spills = '--spills' in sys.argv
    for r in rows:
        if sub and sub[0] not in r['name']:
            continue
        if only_spills and not (r['spill'] or r['scratch']):
            continue
        kern = re.sub(r'^_ZN4gptq\d*_GLOBAL__N_1\d+', '', r['name'])
        kern = re.match(r'[A-Za-z_0-9]+?(?=I[LT])', kern).group(0) if re.match(r'[A-Za-z_0-9]+?(?=I[LT])', kern) else kern[:40]
        print('%-28s <%s>  vgpr %3d agpr %3d scratch %4d spill %3d occ %d lds %d' % (
            kern, ', '.join(v for _, v in r['args']), r['vgprs'], r['agprs'], r['scratch'], r['spill'], r['occ'], r['lds']))
    print('%d kernels, %d with scratch or spills' % (len(rows), sum(1 for r in rows if r['spill'] or r['scratch'])))
