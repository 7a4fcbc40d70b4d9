#!/usr/bin/env python3
"""Audit of the gfx950 ISA (hipcc -S --cuda-device-only output): every s_barrier that can be reached with an LDS WRITE of the same
wave still un-waited-for (no `s_waitcnt ... lgkmcnt(0)` on some path from a ds_write* / ds_*_rtn-less atomic to the barrier).
gfx950 does not wait before a barrier by itself; the compiler normally emits the wait (the workgroup-scope release of
__syncthreads()), but round 6 found one path -- `LDS store; continue;` to a barrier at a loop head -- where it did not
(onepass.hpp, persistent loop).  Forward data-flow over basic blocks, per kernel.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I noisereduce_amd/csrc --cuda-device-only -S noisereduce_amd/csrc/api.hip -o /tmp/api.s
  python tools/audit_barrier_waits.py /tmp/api.s [more .s files]
"""
import re, sys
from collections import defaultdict

def kernels(lines):
    cur, name = None, None
    for ln in lines:
        m = re.match(r'^(_Z\w+):\s', ln)
        if m and cur is None:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            cur.append(ln)
            if ln.strip() == 's_endpgm' or ln.strip().startswith('.Lfunc_end'):
                if ln.strip().startswith('.Lfunc_end'):
                    yield name, cur
                    cur, name = None, None

def audit(name, body):
    # basic blocks: split at labels and after branches
    blocks, labels, cur = [], {}, []
    def flush():
        nonlocal cur
        blocks.append(cur); cur = []
    for ln in body:
        t = ln.strip()
        if not t or t.startswith(';') or t.startswith('.') and not re.match(r'^\.LBB\w+:', t):
            continue
        m = re.match(r'^(\.LBB\w+):', t)
        if m:
            if cur: flush()
            labels[m.group(1)] = len(blocks)
            continue
        cur.append(t.split(';')[0].strip())
        if t.startswith('s_branch') or t.startswith('s_cbranch') or t.startswith('s_endpgm'):
            flush()
    if cur: flush()
    succ = defaultdict(list)
    for i, b in enumerate(blocks):
        last = b[-1] if b else ''
        if last.startswith('s_branch'):
            tgt = last.split()[1]
            if tgt in labels: succ[i].append(labels[tgt])
        elif last.startswith('s_cbranch'):
            tgt = last.split()[1]
            if tgt in labels: succ[i].append(labels[tgt])
            if i + 1 < len(blocks): succ[i].append(i + 1)
        elif last.startswith('s_endpgm'):
            pass
        elif i + 1 < len(blocks):
            succ[i].append(i + 1)
    reads_too = bool(__import__('os').environ.get('AUDIT_READS'))   # AUDIT_READS=1: LDS reads count as well (a later overwrite by another wave)
    is_write0 = lambda t: re.match(r'^ds_(write|add|sub|min|max|and|or|xor|inc|dec|cmpst|wrxchg|append|consume|swizzle)?', t) and t.startswith('ds_') and not t.startswith('ds_read') and not t.startswith('ds_bpermute') and not t.startswith('ds_permute') and not t.startswith('ds_swizzle')
    is_write = lambda t: is_write0(t) or (reads_too and t.startswith('ds_read'))
    clears = lambda t: t.startswith('s_waitcnt') and 'lgkmcnt(0)' in t
    pend_in = [False] * len(blocks)
    found = set()
    work = [0]
    pend_in[0] = False
    seen_state = {}
    while work:
        i = work.pop()
        st = pend_in[i]
        key = (i, st)
        if key in seen_state: continue
        seen_state[key] = 1
        p = st
        for k, t in enumerate(blocks[i]):
            if t.startswith('s_barrier') and p:
                found.add((i, k))
            if clears(t): p = False
            elif is_write(t): p = t
        for s in succ[i]:
            if p and not pend_in[s]:
                pend_in[s] = p
                work.append(s)
            elif (s, pend_in[s]) not in seen_state:
                work.append(s)
    return [(i, k, blocks[i][max(0, k - 3):k + 1]) for i, k in sorted(found)], len(blocks)

bad_total = 0
for path in sys.argv[1:]:
    lines = open(path).read().split('\n')
    n = 0
    for name, body in kernels(lines):
        n += 1
        res, nb = audit(name, body)
        if res:
            bad_total += len(res)
            print('%s: %d barrier(s) reachable with an un-waited LDS write' % (name, len(res)))
            for i, k, ctx in res[:6]:
                print('    block %d: ... %s' % (i, ' | '.join(ctx)))
    print('%s: %d kernels audited' % (path, n))
print('barriers flagged:', bad_total)
