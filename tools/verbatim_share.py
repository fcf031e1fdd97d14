#!/usr/bin/env python
"""Developer tool: share of a file's non-trivial lines (stripped, > 25 characters) that appear verbatim somewhere in the
reference's Python sources -- the measure the round-4 review used for the front-end modules.  Needs /root/reference (this
container only).  Usage: python tools/verbatim_share.py [-v] file.py ..."""
import glob
import os
import sys

REF = '/root/reference/omgtools'


def lines_of(path):
    out = []
    for raw in open(path, errors='replace'):
        s = raw.strip()
        if len(s) > 25:
            out.append(s)
    return out


def main():
    verbose = '-v' in sys.argv
    files = [a for a in sys.argv[1:] if a != '-v']
    ref = set()
    for p in glob.glob(os.path.join(REF, '**', '*.py'), recursive=True):
        ref.update(lines_of(p))
    for f in files:
        mine = lines_of(f)
        hit = [s for s in mine if s in ref]
        print('%-50s %4d lines  %4d verbatim  share %.2f' % (f, len(mine), len(hit), len(hit) / max(1, len(mine))))
        if verbose:
            for s in hit:
                print('      ', s)


if __name__ == '__main__':
    main()
