#!/usr/bin/env python3
"""Static instruction mix per kernel of a device assembly listing (hipcc -S --cuda-device-only):
    python tools/isa_loop_stats.py <file.s> [name substring]
total instructions, 32-bit integer multiplies (quarter rate), f64 instructions, ds_bpermute, DPP moves,
scratch accesses, VGPRs / scratch bytes -- and the same for the innermost big loop (the dependency loop of
the sub-node kernels: the largest 'Loop Header' body)."""
import re, sys

def stats(ins):
    return dict(insts=len(ins),
                intmul=sum(1 for x in ins if re.match(r'v_(mul_lo|mul_hi|mad_u64|mad_i64)', x)),
                f64=sum(1 for x in ins if '_f64' in x),
                bpermute=sum(1 for x in ins if 'bpermute' in x),
                dpp=sum(1 for x in ins if 'dpp' in x),
                vmem=sum(1 for x in ins if re.match(r'(global|buffer|flat)_(load|store|atomic)', x)),
                scratch=sum(1 for x in ins if x.startswith('scratch_')),
                salu=sum(1 for x in ins if x.startswith('s_') and not x.startswith(('s_waitcnt', 's_nop'))))

def main():
    txt = open(sys.argv[1]).read().split('\n')
    sub = sys.argv[2] if len(sys.argv) > 2 else ''
    cur, body, meta = None, [], {}
    for l in txt:
        m = re.match(r'^(_Z\w+):', l)
        if m and cur is None:
            cur, body, meta = m.group(1), [], {}
            continue
        if cur is None:
            continue
        body.append(l)
        m = re.match(r'^; (NumVgprs|ScratchSize|Occupancy): (\d+)', l)
        if m:
            meta[m.group(1)] = int(m.group(2))
        if l.startswith('; Occupancy'):
            if sub in cur:
                ins = [x.strip() for x in body if x.strip() and not x.strip().startswith((';', '.')) and not x.strip().endswith(':')]
                print(cur[:100])
                print('   whole kernel', stats(ins), meta)
                # loops: label lines with "Loop Header"; body = until the back edge to that label
                best = None
                for i, x in enumerate(body):
                    mm = re.match(r'^(\.LBB\d+_\d+):.*Loop Header: Depth=(\d+)', x)
                    if not mm:
                        continue
                    lab = mm.group(1)
                    last = max((j for j in range(i, len(body)) if re.search(r's_cbranch\w+ ' + re.escape(lab) + r'\b|s_branch ' + re.escape(lab) + r'\b', body[j])), default=None)
                    if last is None:
                        continue
                    li = [y.strip() for y in body[i:last + 1] if y.strip() and not y.strip().startswith((';', '.')) and not y.strip().endswith(':')]
                    if best is None or len(li) > best[0]:
                        best = (len(li), mm.group(2), li)
                # the dependency loop is the largest loop that is not the outermost (depth >= 2)
                cands = []
                for i, x in enumerate(body):
                    mm = re.match(r'^(\.LBB\d+_\d+):.*Loop Header: Depth=(\d+)', x)
                    if mm and int(mm.group(2)) >= 2:
                        lab = mm.group(1)
                        last = max((j for j in range(i, len(body)) if re.search(r's_cbranch\w+ ' + re.escape(lab) + r'\b|s_branch ' + re.escape(lab) + r'\b', body[j])), default=None)
                        if last:
                            li = [y.strip() for y in body[i:last + 1] if y.strip() and not y.strip().startswith((';', '.')) and not y.strip().endswith(':')]
                            cands.append((len(li), int(mm.group(2)), li))
                if cands:
                    n, d, li = max(cands)
                    print('   largest inner loop (depth %d)' % d, stats(li))
            cur = None

main()
