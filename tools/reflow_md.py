#!/usr/bin/env python3
"""Re-flow markdown paragraphs and list items to at most 118 columns (tables, code blocks and headers untouched).
    tools/reflow_md.py FILE..."""
import re
import sys
import textwrap

W = 118
ITEM = re.compile(r'^(\s*(?:[-*]|\d+\.)\s+)(.*)$')


def flush(buf, out):
    if not buf:
        return
    m = ITEM.match(buf[0])
    if m:
        ind = m.group(1)
        body = ' '.join([m.group(2)] + [b.strip() for b in buf[1:]])
        out.append(textwrap.fill(body, W, initial_indent=ind, subsequent_indent=' ' * len(ind), break_long_words=False,
                                 break_on_hyphens=False))
    else:
        ind = re.match(r'^(\s*)', buf[0]).group(1)
        body = ' '.join(b.strip() for b in buf)
        out.append(textwrap.fill(body, W, initial_indent=ind, subsequent_indent=ind, break_long_words=False, break_on_hyphens=False))
    buf.clear()


def reflow(text):
    out, buf, code = [], [], False
    for ln in text.split('\n'):
        if ln.startswith('```'):
            flush(buf, out); out.append(ln); code = not code; continue
        if code:
            out.append(ln); continue
        if not ln.strip() or ln.startswith('#') or ln.startswith('|') or ln.startswith('    '):
            flush(buf, out); out.append(ln); continue
        if ITEM.match(ln):
            flush(buf, out)
        buf.append(ln)
    flush(buf, out)
    return '\n'.join(out)


if __name__ == '__main__':
    for fn in sys.argv[1:]:
        s = open(fn).read()
        open(fn, 'w').write(reflow(s))
