#!/usr/bin/env python3
"""Basic blocks of one kernel in an llvm-objdump listing (tools/disasm.sh), with what matters for a hot loop:
instruction mix per block, scratch traffic, waits, and which blocks sit inside backward branches (loops).

    tools/isa_loops.py LISTING KERNEL-SUBSTRING [--min N] [--dump FIRST_BLOCK LAST_BLOCK]

Prints one line per block of >= N instructions (default 12): index, address offset, instruction count, counts of
fp64 VALU (fma/mul/add/max/min/rcp/rsq/cvt), other VALU, SALU, LDS, VMEM, scratch loads / stores, branches, and the
depth of loop nesting (number of backward branches that span the block)."""
import re
import sys


def main():
    lst, pat = sys.argv[1], sys.argv[2]
    minn = 12
    dump = None
    if '--min' in sys.argv:
        minn = int(sys.argv[sys.argv.index('--min') + 1])
    if '--dump' in sys.argv:
        k = sys.argv.index('--dump')
        dump = (int(sys.argv[k + 1]), int(sys.argv[k + 2]))
    lines = open(lst).read().split('\n')
    start = None
    for i, ln in enumerate(lines):
        if ln.endswith('>:') and pat in ln:
            start = i
            break
    if start is None:
        raise SystemExit('kernel not found')
    base = int(lines[start].split()[0], 16)
    ins = []                                              # (offset, text, branch target offset or None)
    for ln in lines[start + 1:]:
        if ln.endswith('>:') or not ln.strip():
            if ln.endswith('>:'):
                break
            continue
        m = re.match(r'\s+(\S.*?)\s+// ([0-9A-F]+):', ln)
        if not m:
            continue
        off = int(m.group(2), 16) - base
        tgt = None
        mt = re.search(r'<[^>]*\+0x([0-9a-f]+)>\s*$', ln)
        if mt and (m.group(1).startswith('s_cbranch') or m.group(1).startswith('s_branch')):
            tgt = int(mt.group(1), 16)
        ins.append((off, m.group(1), tgt))
    leaders = {0}
    for k, (off, tx, tgt) in enumerate(ins):
        if tgt is not None:
            leaders.add(tgt)
            if k + 1 < len(ins):
                leaders.add(ins[k + 1][0])
        if tx.startswith('s_endpgm') and k + 1 < len(ins):
            leaders.add(ins[k + 1][0])
    blocks, cur = [], []
    for it in ins:
        if it[0] in leaders and cur:
            blocks.append(cur)
            cur = []
        cur.append(it)
    if cur:
        blocks.append(cur)
    boff = [b[0][0] for b in blocks]
    back = []                                             # (from offset, to offset) of backward branches
    for b in blocks:
        for off, tx, tgt in b:
            if tgt is not None and tgt <= off:
                back.append((tgt, off))
    print('%d instructions, %d blocks, %d backward branches' % (len(ins), len(blocks), len(back)))
    f64 = re.compile(r'v_(fma|mul|add|max|min|rcp|rsq|cvt|fract|ldexp|frexp|cmp\w*|trig|div|sqrt)\w*_f64|v_cvt_f64|v_cvt_\w+_f64')
    for bi, b in enumerate(blocks):
        n = len(b)
        c = dict(f64=0, valu=0, salu=0, lds=0, vmem=0, sl=0, ss=0, br=0, wait=0)
        for off, tx, tgt in b:
            op = tx.split()[0]
            if op.startswith('scratch_load'):
                c['sl'] += 1
            elif op.startswith('scratch_store'):
                c['ss'] += 1
            elif op.startswith('ds_'):
                c['lds'] += 1
            elif op.startswith(('global_', 'buffer_', 'flat_')):
                c['vmem'] += 1
            elif op.startswith('s_waitcnt'):
                c['wait'] += 1
            elif op.startswith(('s_cbranch', 's_branch')):
                c['br'] += 1
            elif op.startswith('s_'):
                c['salu'] += 1
            elif f64.match(op):
                c['f64'] += 1
            elif op.startswith('v_'):
                c['valu'] += 1
        depth = sum(1 for (a, z) in back if a <= b[0][0] <= z)
        if n >= minn or c['sl'] + c['ss'] > 0 and depth > 0:
            print('B%-4d +0x%05x n=%-4d f64=%-3d valu=%-3d salu=%-3d lds=%-3d vmem=%-2d scr_ld=%-2d scr_st=%-2d wait=%-2d br=%d depth=%d'
                  % (bi, b[0][0], n, c['f64'], c['valu'], c['salu'], c['lds'], c['vmem'], c['sl'], c['ss'], c['wait'], c['br'], depth))
        if dump and dump[0] <= bi <= dump[1]:
            for off, tx, tgt in b:
                print('      +0x%05x  %s%s' % (off, tx, ('   -> +0x%05x' % tgt) if tgt is not None else ''))


if __name__ == '__main__':
    main()
