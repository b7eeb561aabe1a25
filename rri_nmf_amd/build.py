"""Builds librri_hip.so (the C-ABI library of include/rri_hip.h) in-tree for gfx950.

    python -m rri_nmf_amd.build [--report]

hipcc cross-compiles without a GPU.  The .so lands in rri_nmf_amd/lib/ (git-ignored,
but it travels to the GPU box with the working tree).
"""
import os
import re
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
SRC = os.path.join(PKG, 'csrc', 'rri_hip.hip')
DEPS = [SRC] + sorted(os.path.join(PKG, 'csrc', f) for f in os.listdir(os.path.join(PKG, 'csrc')) if f.endswith('.hpp')) + \
    [os.path.join(ROOT, 'include', 'rri_hip.h')]
LIB = os.path.join(PKG, 'lib', 'librri_hip.so')


def hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return 'hipcc'


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    return all(os.path.getmtime(p) <= t for p in DEPS if os.path.exists(p))


def build(force=False, report=False):
    if up_to_date() and not force and not report:
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [hipcc(), '-O3', '-std=c++17', '--offload-arch=gfx950', '-shared', '-fPIC',
           '-I' + os.path.join(ROOT, 'include'), SRC, '-o', LIB]
    if report:
        cmd.append('-Rpass-analysis=kernel-resource-usage')
    print(' '.join(cmd), flush=True)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout)
        raise RuntimeError('hipcc failed building librri_hip.so')
    if report:
        print_report(res.stdout)
    return LIB


def print_report(text):
    rows, cur = [], {}
    for line in text.splitlines():
        m = re.search(r'remark:\s+(.*?) \[-Rpass', line)
        if not m:
            continue
        body = m.group(1).strip()
        if body.startswith('Function Name:'):
            cur = {'name': body.split(':', 1)[1].strip()}
            rows.append(cur)
        elif ':' in body:
            kk, vv = body.split(':', 1)
            cur[kk.strip()] = vv.strip()
    for r in rows:
        name = subprocess.run(['c++filt', r['name']], stdout=subprocess.PIPE, text=True).stdout.strip()
        name = re.sub(r'\(.*', '', name)
        name = name.replace('void ', '').replace('rri::', '')
        print('%-64s vgpr %-4s agpr %-3s sgpr %-4s spill %s/%s  lds %-6s occ %s' % (
            name[-64:], r.get('VGPRs', '?'), r.get('AGPRs', '?'), r.get('TotalSGPRs', '?'),
            r.get('VGPRs Spill', '?'), r.get('SGPRs Spill', '?'), r.get('LDS Size [bytes/block]', '?'),
            r.get('Occupancy [waves/SIMD]', '?')))


if __name__ == '__main__':
    print(build(force=True, report='--report' in sys.argv))
