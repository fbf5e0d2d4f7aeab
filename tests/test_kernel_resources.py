"""CPU (no GPU): register budgets of the workhorse kernels, read from hipcc's resource-usage remarks for gfx950.

A spilled register in one of these kernels is reloaded from scratch inside its tile loop or epilogue, one `s_waitcnt vmcnt(0)` at a
time (round 3: 20-25 spilled VGPRs in the 128-column fp32 forward kernel, caused by the heads' MSE epilogue being compiled into every
forward instantiation, cost 1.6 % of the train step and 4.7 % of the eval forward -- DESIGN.md section 8).  The list is the set of
instantiations the headline configurations launch at 64 x 64 and 32 x 32; `tools/isa_scan.py` prints the whole picture."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'cu_net_amd', 'csrc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-munsafe-fp-atomics', '-fPIC', '-Wno-unused-function', '--cuda-device-only',
         '-Rpass-analysis=kernel-resource-usage', '-c']

# file -> {mangled-name fragment: VGPR budget}  (budget = 512 / waves per SIMD the launcher may place)
EXPECT = {
    'conv_kernels.hip': {
        'conv_kernelILi0ELi0ELi4ELb1ELi0E': 168,           # fp32 1x1 forward, 128 output columns
        'conv_pair_kernelILi0ELi0ELi4ELb1ELi0E': 168,      # the adapter pair of it
        'conv_kernelILi0ELi0ELi3ELb1ELi3E': 168,           # heat-map head (K = 68) with the fused MSE epilogue
        'conv_kernelILi2ELi1ELi1ELb1ELi0E': 168,           # fp32 1x1 data gradient (LDS-tile epilogue)
        'conv_pair_kernelILi2ELi1ELi1ELb1ELi0E': 168,
        'conv_kernelILi3ELi1ELi1ELb1ELi0E': 168,           # fp32 3x3 data gradient
        # the same on the bf16 matrix pipe (planner option f32_split, the default: XBG = 6 / 7)
        'conv_kernelILi0ELi0ELi4ELb1ELi6E': (168, 1),      # (one register spilled at set-up and reloaded once per tile in the epilogue, outside the chunk loop)
        'conv_pair_kernelILi0ELi0ELi4ELb1ELi6E': 168,
        'conv_kernelILi0ELi0ELi3ELb1ELi7E': 168,
        'conv_kernelILi2ELi1ELi1ELb1ELi6E': 128,
        'conv_pair_kernelILi2ELi1ELi1ELb1ELi6E': 128,
        'conv_kernelILi3ELi1ELi1ELb1ELi6E': 128,
        'conv3x3_ring_split_kernel': 256,
        'dgrad3x3_ring_split_kernel': 256,
        'stem_fwd_split_kernel': (256, 6),                 # (six registers spilled before the row loop, reloaded once per output row / at the end: none inside the tile loop)
        'dgrad1x1_rows_split_kernel': 256,
        'dgrad1x1_rows_split2_kernel': 256,                # the dominant kernel of the bench line (round 5; round 6: tiles dealt from a counter)
    },
    'wgrad3_kernels.hip': {
        'wgrad3_kernelILi5ELb0ELi0ELb1E': 256,             # 1x1 weight gradient, 320 channels, split contraction
        'wgrad3_kernelILi5ELb1ELi0ELb1E': 256,
        'wgrad3_3x3_kernelILi0ELb1E': 256,
        'wgrad3_stem_kernelILb1ELb1E': 256,                # stem weight gradient: dz computed while staging, split contraction (round 5)
        'wgrad5_split_kernelILi5E': 256,                   # round 6 (planner option wgrad_split_planes): operands cut once, planes + transpose reads
        'wgrad5_split_kernelILi8E': 256,
        'wgrad3_stem_planes_kernelILb1E': 256,             # round 6 (planner option stem_wgrad_planes, default): ring of bf16 planes, consumer / producer waves
        'wgrad3_stem_planes_kernelILb0E': 256,
    },
    'bf16_kernels.hip': {
        'dgrad_bf16_kernelILi1ELi2ELi4ELb0E': 168,         # bf16 1x1 data gradient, two channel tiles, K = 128 (12 waves)
        'dgrad_bf16_pair_kernelILi1ELi2ELi4E': 168,
        'dgrad_bf16_kernelILi1ELi1ELi4ELb0E': 128,
        'dgrad_bf16_kernelILi1ELi2ELi4ELb1E': 256,         # round 6, planner option fuse_z_gather (off by default): the gather folded into the load, 8 / 12 waves
        'dgrad_bf16_kernelILi1ELi1ELi4ELb1E': 168,
        'conv_bf16_kernelILi1ELi1ELi0E': 128,              # bf16 1x1 forward, one / two channel tiles
        'conv_bf16_kernelILi1ELi2ELi0E': 128,
    },
}


@pytest.mark.parametrize('fname', sorted(EXPECT))
def test_workhorse_kernels_do_not_spill(fname, tmp_path):
    if shutil.which('hipcc') is None:
        pytest.skip('hipcc not on PATH')
    r = subprocess.run(['hipcc'] + FLAGS + [os.path.join(SRC, fname), '-o', str(tmp_path / 'k.o')], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            cur = m.group(1)
            res[cur] = {}
        m = re.search(r'remark:\s+(VGPRs|VGPRs Spill|SGPRs Spill): (\d+)', line)
        if m and cur:
            res[cur][m.group(1)] = int(m.group(2))
    for frag, budget in EXPECT[fname].items():
        budget, max_spill = budget if isinstance(budget, tuple) else (budget, 0)
        hits = {k: v for k, v in res.items() if frag in k}
        assert hits, f'{frag}: no such instantiation in {fname}'
        for k, v in hits.items():
            assert v.get('VGPRs Spill', 0) <= max_spill and v.get('SGPRs Spill', 0) == 0, (k, v)
            assert v['VGPRs'] <= budget, (k, v, budget)


def _vgprs(text):
    """Indices of the VGPRs an operand string names: v7, v[22:25] ..."""
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]', text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'\bv(\d+)\b', text):
        out.add(int(m.group(1)))
    return out


def test_row_tile_data_gradient_keeps_its_uncounted_requests_untouched(tmp_path):
    """dgrad1x1_rows_split2_kernel (the dominant kernel of the bench line) requests its x pieces by inline asm -- loads the compiler's
    vmcnt bookkeeping does not see -- and waits for them at the top of the NEXT iteration with a hand-counted `s_waitcnt vmcnt(4)`
    (everything but the previous tile's four dz stores).  That is only correct while (a) the four requested register quads are neither
    read, written nor copied between the request and the wait, across the loop's back edge, (b) exactly four global stores follow the
    last request of an iteration, and (c) nothing of the kernel lives in scratch.  A compiler or flag change that breaks any of them would
    produce stale ReLU masks / BatchNorm sums silently (round-5 advice): this reads the gfx950 ISA and fails instead."""
    if shutil.which('hipcc') is None:
        pytest.skip('hipcc not on PATH')
    asm = tmp_path / 'conv.s'
    flags = [f for f in FLAGS if not f.startswith('-Rpass') and f != '-c']
    r = subprocess.run(['hipcc'] + flags + ['-S', os.path.join(SRC, 'conv_kernels.hip'), '-o', str(asm)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = asm.read_text().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r'_ZN5cunet27dgrad1x1_rows_split2_kernel\w*:', l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
    body = [l.split(';')[0].strip() for l in lines[start:end]]
    assert not any(l.startswith('scratch_') for l in body), 'the kernel spills to scratch'
    # hipcc lays basic blocks out in any order: the checks walk the control flow (fall-through, s_branch, both arms of s_cbranch_*) instead
    # of reading the listing top to bottom
    labels = {l[:-1]: i for i, l in enumerate(body) if re.match(r'\.LBB\d+_\d+:$', l)}
    headers = [i for i in range(start, end) if re.match(r'\.LBB\d+_\d+:', lines[i]) and 'Loop Header: Depth=1' in lines[i]]
    w4 = [i for i, l in enumerate(body) if l == 's_waitcnt vmcnt(4)']
    assert 1 <= len(w4) <= 2, w4
    # the x requests' wait is the first vmcnt(4) behind the header of the tile loop (the other one, round 6, is the tile ticket's: behind the stores)
    tops = [h - start for h in headers if any(h - start < w for w in w4)]
    assert tops, 'no loop header in front of the counted wait'
    top = max(t for t in tops if t < min(w for w in w4 if w > min(tops)))
    wx = min(w for w in w4 if w > top)
    # (the loop's top has three arms -- the counted wait; idle waves, which wait for their DMA pieces with vmcnt(0) and never read the x
    # registers; the workgroup's first tile, whose requests were waited for in front of the loop -- and they meet in front of the barrier: a walk
    # ends there)
    stops = {min(i for i in range(wx, len(body)) if body[i] == 's_barrier')}

    def walk(frm):
        """instructions executed after `frm` on any path, up to (not including) the barrier behind the x requests' wait"""
        seen, todo = set(), [frm + 1]
        while todo:
            i = todo.pop()
            while i < len(body) and i not in seen and i not in stops:
                seen.add(i)
                l = body[i]
                m = re.match(r's_branch\s+(\.LBB\d+_\d+)$', l)
                if m:
                    i = labels[m.group(1)]
                    continue
                m = re.match(r's_cbranch\w*\s+(\.LBB\d+_\d+)$', l)
                if m:
                    todo.append(labels[m.group(1)])
                if l.startswith('s_endpgm'):
                    break
                i += 1
        return seen

    # blocks of the tile loop, by hipcc's own annotation of every block ("in Loop: Header=BBn_m"; the header says "This Loop Header")
    hdr = re.match(r'\.(LBB\d+_\d+):', lines[start + top]).group(1)
    inloop, flag = [], False
    for raw in lines[start:end]:
        if re.match(r'\.LBB\d+_\d+:', raw) or re.match(r';\s*%bb\.\d+:', raw):
            flag = f'Header={hdr[1:]}' in raw or raw.startswith(f'.{hdr}:')
        inloop.append(flag)
    after_wait = {i for i in walk(min(stops)) if inloop[i]}      # one iteration: from the barrier round to the barrier again
    loads = [(i, body[i]) for i in sorted(after_wait) if body[i].startswith('global_load_dwordx4')]
    stores = [i for i in sorted(after_wait) if body[i].startswith('global_store_dwordx4')]
    assert len(loads) == 4 and len(stores) == 4, (len(loads), len(stores))
    for st in stores:
        assert not any(body[j].startswith('global_load_dwordx4') for j in walk(st)), 'an x request behind a dz store: vmcnt(4) would no longer cover the requests'
    others = [body[i] for i in after_wait if body[i].startswith(('global_load', 'global_atomic', 'buffer_load', 'buffer_store', 'global_store'))
              and not body[i].startswith(('global_load_dwordx4', 'global_store_dwordx4', 'global_load_lds'))]
    # (round 6: thread 0 takes the workgroup's next tile from a counter -- ONE returning atomic per iteration, issued in front of the x requests,
    # i.e. older than the four dz stores the counted wait leaves in flight)
    assert all(o.startswith('global_atomic_add ') for o in others) and len(others) <= 1, others
    for i, l in enumerate(body):
        if l.startswith('global_atomic_add ') and i in after_wait:
            assert all(j in walk(i) for j, _ in loads), 'the tile ticket is not older than the x requests'
    for i, l in loads + [(i, body[i]) for i in after_wait if body[i].startswith('global_atomic_add ')]:
        dst = _vgprs(l.split(',')[0])
        assert len(dst) in (1, 4), l
        if len(dst) == 1:
            continue                                             # (the ticket's register is read behind ITS wait, checked by the stores' order above)
        for j in walk(i):
            ops = body[j].split(None, 1)
            if len(ops) == 2 and dst & _vgprs(ops[1]):
                raise AssertionError(f'{body[j]!r} touches the destination of the in-flight request {l!r}')
