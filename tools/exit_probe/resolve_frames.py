"""Resolves the raw frame addresses of a glog / rocprofv3 fault report against a /proc/self/maps dump of the SAME process:
prints, per frame, the DSO, the offset in it and -- when the file exists in this container too (same image as the GPU box) --
the nearest dynamic symbol below the offset.

    python3 tools/exit_probe/resolve_frames.py <log-with-frames> <maps-file>
"""
import bisect
import re
import subprocess
import sys


def load_maps(path):
    segs = []
    for line in open(path):
        m = re.match(r'([0-9a-f]+)-([0-9a-f]+) (\S+) ([0-9a-f]+) \S+ \d+\s*(.*)', line)
        if m:
            segs.append((int(m.group(1), 16), int(m.group(2), 16), m.group(3), int(m.group(4), 16), m.group(5).strip()))
    return segs


_syms = {}


def symbols(dso):
    if dso not in _syms:
        tab = []
        try:
            out = subprocess.run(['nm', '-D', '-C', '--defined-only', dso], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                 text=True).stdout
            out += subprocess.run(['nm', '-C', '--defined-only', dso], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                  text=True).stdout
            for ln in out.splitlines():
                p = ln.split(' ', 2)
                if len(p) == 3 and p[1] in 'TtWwiV':
                    tab.append((int(p[0], 16), p[2]))
        except OSError:
            pass
        tab.sort()
        _syms[dso] = tab
    return _syms[dso]


def resolve(addr, segs):
    for lo, hi, perm, off, name in segs:
        if lo <= addr < hi:
            base = min(s[0] - s[3] for s in segs if s[4] == name) if name else lo
            rel = addr - base
            sym = ''
            if name.startswith('/'):
                tab = symbols(name)
                i = bisect.bisect_right([a for a, _ in tab], rel) - 1
                if i >= 0:
                    sym = '%s+0x%x' % (tab[i][1], rel - tab[i][0])
            return name or '[anon]', rel, perm, sym
    return '[unmapped]', addr, '', ''


if __name__ == '__main__':
    log, maps = sys.argv[1], sys.argv[2]
    segs = load_maps(maps)
    for line in open(log, errors='replace'):
        m = re.search(r'(?:@|PC: @)\s+0x([0-9a-f]+)\s+(\S.*)?$', line.rstrip())
        if not m:
            m2 = re.search(r'SIGSEGV \(@0x([0-9a-f]+)\)', line)
            if m2:
                print('fault address 0x%s -> %s' % (m2.group(1), resolve(int(m2.group(1), 16), segs)[:3]))
            continue
        addr = int(m.group(1), 16)
        dso, rel, perm, sym = resolve(addr, segs)
        print('0x%x  %-60s +0x%-8x %s  %s' % (addr, dso, rel, perm, sym or (m.group(2) or '')))
