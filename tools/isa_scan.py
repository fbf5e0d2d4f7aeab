#!/usr/bin/env python3
"""Static scan of the gfx950 ISA of every kernel in cu_net_amd/csrc (no GPU needed; tools only).

    python tools/isa_scan.py [--min-serial 8]

For each kernel: VGPRs and spilled VGPRs (hipcc -Rpass-analysis=kernel-resource-usage) and the number of memory requests that
are waited for ONE BY ONE -- a `global_load` / `scratch_load` or `ds_read` whose own `s_waitcnt vmcnt(0)` / `lgkmcnt(0)` follows
within a few instructions with no other request in between.  Each of those is a full round trip the wave sits out.  What this
found in round 3: the element-wise LDS epilogues of the data gradients (16 dependent LDS round trips per tile), the look-ahead
requests behind run-time branches (the compiler falls back to vmcnt(0)), and 20-25 spilled registers in the 128-column fp32
forward kernel that were reloaded from scratch one at a time in every tile's store epilogue.
Prints kernels with spills, then kernels with at least --min-serial serialised requests."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'cu_net_amd', 'csrc')
FILES = ['conv_kernels', 'wgrad_kernels', 'wgrad3_kernels', 'elementwise_kernels', 'quant_kernels', 'bf16_kernels', 'augment_kernels']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-munsafe-fp-atomics', '-fPIC', '-Wno-unused-function']


def demangle(name):
    try:
        return subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().replace('cunet::', '')
    except OSError:
        return name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--min-serial', type=int, default=8)
    a = ap.parse_args()
    spills, serial = [], []
    with tempfile.TemporaryDirectory() as tmp:
        for f in FILES:
            src = os.path.join(SRC, f + '.hip')
            asm = os.path.join(tmp, f + '.s')
            r = subprocess.run(['hipcc'] + FLAGS + ['--cuda-device-only', '-S', '-Rpass-analysis=kernel-resource-usage', '-o', asm, src],
                               capture_output=True, text=True)
            if r.returncode != 0:
                sys.exit(r.stderr[-2000:])
            cur = None
            res = {}
            for line in r.stderr.splitlines():
                m = re.search(r'Function Name: (\S+)', line)
                if m:
                    cur = m.group(1)
                    res[cur] = {}
                m = re.search(r'remark:\s+(VGPRs|VGPRs Spill|ScratchSize \[bytes/lane\]): (\d+)', line)
                if m and cur:
                    res[cur][m.group(1)] = int(m.group(2))
            for k, v in res.items():
                if v.get('VGPRs Spill', 0) > 0:
                    spills.append((f, demangle(k), v.get('VGPRs'), v['VGPRs Spill'], v.get('ScratchSize [bytes/lane]')))
            txt = open(asm).read().split('\n')
            cur = None
            st = {}
            for i, line in enumerate(txt):
                m = re.match(r'^(_ZN5cunet\w+):', line)
                if m:
                    cur = m.group(1)
                    st[cur] = [0, 0, 0]          # serialised vector-memory loads, serialised LDS reads, MFMAs
                    continue
                if cur is None:
                    continue
                if 'v_mfma' in line:
                    st[cur][2] += 1
                if re.search(r'\b(global_load|buffer_load|scratch_load)', line):
                    nxt = txt[i + 1:i + 5]
                    if any('s_waitcnt vmcnt(0)' in x for x in nxt) and not any(re.search(r'global_load|buffer_load|scratch_load', x) for x in nxt):
                        st[cur][0] += 1
                if re.search(r'\bds_read', line):
                    nxt = txt[i + 1:i + 4]
                    if any(re.search(r's_waitcnt.*lgkmcnt\(0\)', x) for x in nxt) and not any('ds_read' in x for x in nxt):
                        st[cur][1] += 1
            for k, (ld, lds, mf) in st.items():
                if ld + lds >= a.min_serial:
                    serial.append((f, demangle(k), ld, lds, mf))
    print('kernels with spilled VGPRs (file, kernel, VGPRs, spilled, scratch bytes per lane):')
    for s in spills:
        print('  %-20s %-60s %4s %4s %5s' % s)
    print('kernels with serialised requests (file, kernel, vector-memory loads, LDS reads, MFMAs in the listing):')
    for s in sorted(serial, key=lambda t: -(t[2] + t[3])):
        print('  %-20s %-60s %4d %4d %5d' % s)


if __name__ == '__main__':
    main()
