#!/usr/bin/env python3
"""One-off of round 5: split the single 118 KB DESIGN.md (rounds 1-4, lines of up to 1 000 characters) into per-kernel files under
docs/ with lines of at most 120 columns.  Paragraphs and list items are re-wrapped; table rows whose cells are long become
"**first cell** -- second cell -- ..." paragraphs.  Kept for the record of how docs/ was produced; not part of the product."""
import os
import re
import sys
import textwrap

W = 118


def wrap_par(line):
    m = re.match(r'^(\s*(?:[-*]|\d+\.)\s+)(.*)$', line)
    if m:
        ind, body = m.group(1), m.group(2)
        return textwrap.fill(body, W, initial_indent=ind, subsequent_indent=' ' * len(ind), break_long_words=False, break_on_hyphens=False)
    m = re.match(r'^(\s*)(.*)$', line)
    ind, body = m.group(1), m.group(2)
    return textwrap.fill(body, W, initial_indent=ind, subsequent_indent=ind, break_long_words=False, break_on_hyphens=False)


def convert(text):
    out = []
    lines = text.split('\n')
    k = 0
    while k < len(lines):
        ln = lines[k]
        if ln.startswith('|'):
            tbl = []
            while k < len(lines) and lines[k].startswith('|'):
                tbl.append(lines[k]); k += 1
            if max(len(r) for r in tbl) <= 120:
                out.extend(tbl)
            else:
                rows = [[c.strip() for c in r.strip().strip('|').split('|')] for r in tbl]
                head = rows[0]
                for r in rows[2:]:
                    if not any(r):
                        continue
                    first = '**' + r[0] + '**' if r[0] else ''
                    parts = []
                    for h, c in zip(head[1:], r[1:]):
                        if c:
                            parts.append(('*' + h + ':* ' if h else '') + c)
                    out.append(wrap_par('- ' + first + (' -- ' if first and parts else '') + '  '.join(parts)))
                out.append('')
            continue
        if ln.startswith('```'):
            out.append(ln); k += 1
            while k < len(lines) and not lines[k].startswith('```'):
                out.append(lines[k]); k += 1
            if k < len(lines):
                out.append(lines[k]); k += 1
            continue
        if len(ln) <= 120 or ln.startswith('#'):
            out.append(ln if not ln.startswith('#') or len(ln) <= 120 else ln[:117] + '...')
        else:
            out.append(wrap_par(ln))
        k += 1
    return '\n'.join(out)


def sections(text):
    """[(level, title, body)] split at ## / ### headers"""
    res, cur = [], None
    for ln in text.split('\n'):
        m = re.match(r'^(#{2,3}) (.*)$', ln)
        if m:
            if cur:
                res.append(cur)
            cur = [len(m.group(1)), m.group(2), []]
        elif cur:
            cur[2].append(ln)
    if cur:
        res.append(cur)
    return [(a, b, '\n'.join(c)) for a, b, c in res]


if __name__ == '__main__':
    src, outdir = sys.argv[1], sys.argv[2]
    secs = sections(open(src).read())
    plan = {  # file -> (title, [section-title prefixes])
        'k_hist.md': ('k_hist -- the one-pass histogram kernel (fit_predict, headline)', ['3.1 ']),
        'k_fused.md': ('k_fused -- the round-2 single-pass kernel (ln-priors, grid KDE, many dictionary widths, wild values)', ['3.1b']),
        'planes_predict.md': ('k_planes / k_plane_rows / k_plane_fused -- fit() planes and predict() from a stored plane', ['3.2 ', '3.3 ']),
        'modec.md': ('Mode C -- free scale with model errors', ['3.4 ']),
        'knn.md': ('k-NN -- the Monte-Carlo nearest-neighbour search and its subset likelihood', ['3.5 ']),
        'arithmetic.md': ('Arithmetic (fast math, MFMA question, removed decompositions)', ['3.6 ', '3.7 ']),
        'summary_nz.md': ('pdfs_summarize and the n(z) steps', ['3.8 ']),
        'measurement_r3_r4.md': ('Measurement, rounds 3-4 (superseded by docs/measurement.md where they overlap)', ['4. ']),
        'multi_gpu.md': ('Multi-GPU', ['5. ']),
        'deviations.md': ('Deliberate deviations and what is out of scope', ['6. ', '7. ']),
        'headroom_r4.md': ('Known headroom as of round 4', ['8. ']),
    }
    for fn, (title, prefs) in plan.items():
        body = ['# ' + title, '', '(split out of the rounds 1-4 DESIGN.md by tools/split_design.py; text unchanged except for line wrapping)', '']
        for lvl, t, b in secs:
            if any(t.startswith(p) for p in prefs):
                body.append('## ' + (t if len(t) <= 110 else t[:107] + '...'))
                body.append(convert(b))
        open(os.path.join(outdir, fn), 'w').write('\n'.join(body).rstrip() + '\n')
    # history: section 0
    hist = ['# History: what earlier rounds did (superseded state)', '']
    for lvl, t, b in secs:
        if t.startswith('0. '):
            hist.append('## Round 4 at a glance (from the round-4 DESIGN.md)'); hist.append(convert(b))
    open(os.path.join(outdir, '..', 'profiles', 'HISTORY.md'), 'w').write('\n'.join(hist).rstrip() + '\n')
    for lvl, t, b in secs:
        if t.startswith('1. ') or t.startswith('2. '):
            open(os.path.join(outdir, '_sec%s.md' % t[0]), 'w').write(convert(b))
