#!/usr/bin/env python3
"""PPO update-steps/sec of the MI355X-native learner hot path (BASELINE.json metric).

One *step* = one PPO update-step = one policy minibatch step + one value minibatch step on a
per-GPU minibatch of 256 x (4 frames of 90x120x3 + road/vehicle/navigation vectors), fp32:
2 x (train-mode CARLANetwork trunk fwd+bwd), 2 trunk Adam steps, policy head fwd/bwd + Beta-PPO
loss + per-tensor clip + old-policy copy + Adam, value head likewise.
Inputs are resident in HBM when the timed region starts.  N>1: one process per GPU (torchrun),
rollout buffer sharded by env shard, one RCCL all-reduce of the fused gradient arena per pass.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# per-frame sum over {conv3x3, 1x1 conv, dw3x3, maxpool, GAP} of (in + out) elements
ALG_ELEMS_PER_FRAME = {(90, 120): 979500, (90, 360): 2902212, (135, 180): 2275248, (135, 540): 6645476}
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def alg_bytes_per_update_step(B, T, H, W, elem_bytes=4):
    """ALG_BYTES_pass = 3 * sizeof(dtype) * B * T * sum(in+out); an update-step is 2 passes (24.07 GB at
    B=256 float32, 48.14 GB at B=1024 bf16)."""
    return 2 * 3 * elem_bytes * B * T * ALG_ELEMS_PER_FRAME[(H, W)]


def pmc_traffic(B, T, H, W, dtype='f32'):
    """HBM bytes per update-step from the committed PMC run of THIS workload and storage type (profiles/*_pmc_traffic.json;
    collected and corrected as MI355X_MICROARCH.md prescribes, see the file's `method`), or None if no run matches."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.json'))):
        try:
            d = json.load(open(f))
            w = d['workload']
            if (w['B'], w['T'], w['H'], w['W'], w.get('dtype', 'f32')) == (B, T, H, W, dtype):
                best = d['bytes_per_update_step']
        except Exception:
            pass
    return best


def pmc_mfma(ms_per_step):
    """MFMA figures of the committed PMC run (profiles/*_pmc_mfma.json, tools/pmc_mfma.sh): utilisation of the float32 GEMM
    kernels and the float32 matrix-pipe rate over the update-step against the 157.3 TFLOP/s peak."""
    import glob
    import re
    # profiles/rNN_pmc_mfma.json of the benchmark workload only (the configuration-3 runs are named r02_c3_*)
    files = sorted(f for f in glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_mfma.json')) if re.fullmatch(r'r\d+_pmc_mfma\.json', os.path.basename(f)))
    if not files:
        return None
    d = json.load(open(files[-1]))
    utils = [k['mfma_util'] for k in d['kernels'] if k.get('mfma_util')]
    flops = d.get('mfma_flops_issued_per_update_step')
    flops16 = d.get('mfma_bf16_flops_issued_per_update_step')      # split-precision kernels: 6 bf16 products per float32 product
    return dict(source=os.path.relpath(files[-1], ROOT), mfma_util_all_kernel_time=d.get('mfma_util_all_kernels'),
                mfma_util_gemm_kernels=[min(utils), max(utils)] if utils else None, f32_flops_issued_per_update_step=flops,
                flops_note='counter products (SQ_INSTS_VALU_MFMA_MOPS_* x 512 of the profiled run): what was ISSUED to the matrix pipes, '
                           'padding included and split-precision kernels counted on the bf16 pipe -- not the model\'s algorithmic FLOPs',
                f32_mfma_tflops=round(flops / (ms_per_step * 1e-3) / 1e12, 2) if flops else None, f32_mfma_peak_tflops=157.3,
                bf16_flops_issued_per_update_step=flops16,
                bf16_mfma_tflops=round(flops16 / (ms_per_step * 1e-3) / 1e12, 2) if flops16 else None, bf16_mfma_peak_tflops=2500.0)


def dominant_kernel(B, T, H, W):
    """The kernel FAMILY with the largest total time in the committed kernel trace of the benchmark workload
    (profiles/*_kernel_trace_summary.md): a kernel and the partial reducer that only exists because of it are one family (the fused
    conv backward `pwb_kernel<...>` + `pwb_reduce_kernel`: the reducer has no algorithmic bytes of its own), every other kernel name is
    its own family.  Reported: the family's total time per update-step over both streams, its launches, the average duration of one
    (kernel + reducer) pair, the algorithmic bytes of that pair and the fraction of the HBM peak."""
    import glob
    import re
    if (B, T, H, W) != (256, 4, 90, 120):
        return None
    # profiles/rNN_kernel_trace_summary.md = the trace of THIS workload at the round's final commit (other traces carry a tag)
    files = sorted(f for f in glob.glob(os.path.join(ROOT, 'profiles', '*_kernel_trace_summary.md'))
                   if re.fullmatch(r'r\d+_kernel_trace_summary\.md', os.path.basename(f)))
    if not files:
        return None
    rows = []
    for line in open(files[-1]):
        m = re.match(r'\| (\S.*?) \| (\d+) \| ([\d.]+) \| ([\d.]+) \|', line)
        if m and not m.group(1).startswith('kernel'):
            rows.append((m.group(1), int(m.group(2)), float(m.group(3)), float(m.group(4))))
    if not rows:
        return None
    fam_of = lambda n: 'pwb' if n.startswith('pwb_kernel') or n.startswith('pwb_reduce_kernel') else n
    fams = {}
    for n, calls, tot, avg in rows:
        f = fams.setdefault(fam_of(n), dict(total_ms=0.0, members=[]))
        f['total_ms'] += tot
        f['members'].append((n, calls, tot, avg))
    top = max(fams, key=lambda k: fams[k]['total_ms'])
    if top == 'pwb':
        mem = fams[top]['members']
        main_calls = sum(c for n, c, _, _ in mem if n.startswith('pwb_kernel'))
        red_calls = sum(c for n, c, _, _ in mem if n.startswith('pwb_reduce_kernel'))
        # the 24 fused conv backwards of one pass (stages 0 / 1: 4 + 8 units, two convs each): reads dz, y, a and writes da -> 4 M (2 N + 2 K) bytes each
        px0, px1, pxp = 11 * 15, 6 * 8, 22 * 30
        shapes = [(pxp, 24, 58), (px0, 58, 92)] + [(px0, 58, 58)] * 6 + [(px0, 116, 116), (px1, 116, 116)] + [(px1, 116, 116)] * 14
        by = sum(4.0 * B * T * px * (2 * n + 2 * k) for px, k, n in shapes) / len(shapes)
        steps = main_calls / (2.0 * len(shapes))              # update-steps covered by the trace (two passes each)
        pair_us = fams[top]['total_ms'] * 1e3 / main_calls
        out = dict(source=os.path.relpath(files[-1], ROOT), kernel='pwb_kernel<*> + pwb_reduce_kernel (fused conv backward of the stage-0 / stage-1 unit convs)',
                   family_total_ms_per_update_step=round(fams[top]['total_ms'] / steps, 3), launches_per_update_step=int(round(main_calls / steps)),
                   reducer_launches_per_update_step=int(round(red_calls / steps)), avg_us=round(pair_us, 1),
                   algorithmic_bytes_per_launch=by, achieved_GBs=round(by / (pair_us * 1e-6) / 1e9, 1),
                   frac=round(by / (pair_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                   members=[dict(kernel=n, calls_per_update_step=round(c / steps, 1), avg_us=a) for n, c, _, a in sorted(mem, key=lambda r: -r[2])],
                   note='kernel + its partial-tile reducer as ONE family, both streams; avg_us = family time / main-kernel launches; '
                        'algorithmic bytes = mean of 4 M (2 N + 2 K) over the 24 conv backwards of a pass (dz, y, a read, da written)')
        crit = critical_stream_top_kernel()
        if crit:
            out['critical_stream_top_kernel'] = crit
        return out
    name, _, _, avg_us = max(fams[top]['members'], key=lambda r: r[2])
    out = dict(source=os.path.relpath(files[-1], ROOT), kernel=name, avg_us=avg_us)
    crit = critical_stream_top_kernel()
    if crit:
        out['critical_stream_top_kernel'] = crit
    if name.startswith('dwf_bwd_kernel<1, 2, true'):
        # fused depthwise backward of the stride-1 units (BN2-backward apply on load -> filter / bias gradient partials + transposed
        # conv + ReLU6 mask + BN1-backward sums): reads the output gradient, the raw depthwise output and the raw conv-1 output, writes
        # the masked input gradient -- four tensors of frames x pixels x channels floats.  One pass: 3 stage-0 units (11x15x58), 7
        # stage-1 units (6x8x116), 3 stage-2 units (3x4x232)
        shapes = [(165, 58)] * 3 + [(48, 116)] * 7 + [(12, 232)] * 3
        by = sum(4.0 * 4.0 * B * T * px * c for px, c in shapes) / len(shapes)
        out.update(algorithmic_bytes_per_launch=by, achieved_GBs=round(by / (avg_us * 1e-6) / 1e9, 1),
                   frac=round(by / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                   note='critical-stream kernel; average over its 13 launches per pass (three resolutions)')
    elif name.startswith('pwb_kernel<128, 128'):
        # fused conv backward at stage 1 (K = N = 116, M = B*T*48 rows; the stride-2 unit's first conv runs at B*T*165): reads dz, y, a,
        # writes da
        by = 4.0 * 4.0 * B * T * 48 * 116
        out.update(algorithmic_bytes_per_launch=by, achieved_GBs=round(by / (avg_us * 1e-6) / 1e9, 1),
                   frac=round(by / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), note='critical-stream kernel (stage-1 shape)')
    elif name.startswith('pwb_reduce_kernel'):
        # fixed-order reduce of the fused conv backward's per-workgroup partial tiles: reads 256 tiles of KP x NP floats (64 KB each at
        # KP = NP = 128: stage 1; 16 KB at 64 x 64: stage 0) + the column-sum rows, writes dW / db (and BN2's coefficients).  Averaged
        # over the 23 launches of a pass: 16 at stage 1, 7 at stage 0
        by = (16 * 256 * 128 * 128 * 4.0 + 7 * 256 * 64 * 64 * 4.0) / 23
        out.update(algorithmic_bytes_per_launch=by, achieved_GBs=round(by / (avg_us * 1e-6) / 1e9, 1),
                   frac=round(by / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                   note='largest total over BOTH streams (46 launches per update-step: 24 on the critical stream, 22 on the side stream next '
                        'to it); the partial tiles are L2 / Infinity-Cache resident writes of the kernel in front')
    if name.startswith('tn_direct_tr_kernel<4'):
        if name.startswith('tn_direct_tr_kernel<4, false'):
            # pw1 filter gradients (A = the unit's input, D = dz of BN1 recomputed in the operand prologue) of the stage-1 units
            # (K = N = 116: 7 stride-1 units at M = B*T*48 rows and the stride-2 unit, whose pw1 runs at the INPUT resolution: B*T*180 rows)
            # and of the stage-2 units (K = N = 232, two 128-column blocks: 3 stride-1 units at B*T*12 rows + the stride-2 unit at B*T*48)
            shapes = [(184320, 116, 116)] + [(49152, 116, 116)] * 7 + [(49152, 232, 232)] + [(12288, 232, 232)] * 3
        else:
            # pw2 filter gradients (A = BN2-applied depthwise output through the A prologue, D = dz of BN3): every unit at its OUTPUT
            # resolution -- 8 stage-1 units at B*T*48 rows (K = N = 116), 4 stage-2 units at B*T*12 rows (K = N = 232)
            shapes = [(49152, 116, 116)] * 8 + [(12288, 232, 232)] * 4
        by = sum(4.0 * m_ * (k + n) for m_, k, n in shapes) / len(shapes)
        out.update(algorithmic_bytes_per_launch=by, achieved_GBs=round(by / (avg_us * 1e-6) / 1e9, 1),
                   frac=round(by / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                   note='side-stream kernel: its duration in the step includes sharing the CUs with the critical stream')
    return out


def critical_stream_top_kernel():
    """The kernel with the largest total time ON THE CRITICAL STREAM of one update-step, from the committed per-stream timeline
    (profiles/rNN_timeline_step.txt, tools/timeline_step.py): the dominant kernel by total time may sit on the side stream."""
    import glob
    import re
    files = sorted(f for f in glob.glob(os.path.join(ROOT, 'profiles', '*_timeline_step.txt'))
                   if re.fullmatch(r'r\d+_timeline_step\.txt', os.path.basename(f)))
    if not files:
        return None
    # format: "queue Q: N kernels, busy X ms" followed by "  total ms   launches   avg us  kernel" lines; the critical stream is the
    # queue with the largest busy time
    queues, cur = [], None
    for line in open(files[-1]):
        m = re.match(r'queue (\d+): (\d+) kernels, busy ([\d.]+) ms', line)
        if m:
            cur = dict(queue=int(m.group(1)), kernels=int(m.group(2)), busy_ms=float(m.group(3)), rows=[])
            queues.append(cur)
            continue
        m = re.match(r'\s*([\d.]+) ms\s+(\d+)\s+([\d.]+) us\s+(\S.*)$', line)
        if m and cur is not None:
            cur['rows'].append((float(m.group(1)), int(m.group(2)), float(m.group(3)), m.group(4).strip()))
    best = None
    if queues:
        crit = max(queues, key=lambda q: q['busy_ms'])
        if crit['rows']:
            tot, n, avg, name = max(crit['rows'])
            best = dict(kernel=name, total_ms_per_step=tot, launches_per_step=n, avg_us=avg, critical_stream_kernels_per_step=crit['kernels'],
                        critical_stream_busy_ms=crit['busy_ms'])
    if best:
        best['source'] = os.path.relpath(files[-1], ROOT)
    return best


def kernel_rooflines(B, T, nsets=8):
    """Isolated roofline points of the three kernel families that carry the tower (stage-1 shapes of the benchmark
    workload: B*T frames of 6x8 pixels, 116 channels per branch), timed with HIP events on the launch stream through the
    C ABI.  achieved = algorithmic bytes of the launch / average duration of back-to-back launches (includes the launch
    gap).  The launches ROTATE over `nsets` disjoint buffer sets (8 x 45.6 MB = 365 MB > the 256 MB Infinity Cache), so every
    launch streams its operands from HBM: these are HBM figures, not cache-resident ones."""
    import ctypes as C
    import torch
    from carla_driving_rl_agent_amd import _lib
    lib = _lib.load()
    dev = torch.device('cuda', torch.cuda.current_device())
    S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None

    def timeit(fn, iters=32):
        for k in range(nsets):
            fn(k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(iters):
            fn(k % nsets)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e-3

    out = []
    G, px, Cc = T, 48, 116
    Mg = B * px
    M = G * Mg
    a = [torch.randn(M, Cc, device=dev) for _ in range(nsets)]
    y = [torch.empty(M, Cc, device=dev) for _ in range(nsets)]
    w = torch.randn(Cc, Cc, device=dev)
    bias = torch.randn(Cc, device=dev)
    nb = int(lib.cdrl_pwconv_fused_partial_rows(G, Mg, Cc, Cc))
    part = torch.zeros(G * nb * 2 * Cc, dtype=torch.float64, device=dev)
    cold = f'cold: launches rotate over {nsets} buffer sets ({nsets * 2 * M * Cc * 4 / 1e6:.0f} MB > 256 MB Infinity Cache)'
    t = timeit(lambda k: lib.cdrl_pwconv_fused(P(a[k]), Cc, 0, None, P(w), Cc, 1, P(bias), P(y[k]), Cc, 0, 0, G, Mg, Cc, Cc, 1, None,
                                               None, P(part), S()))
    by = 4.0 * M * 2 * Cc
    out.append(dict(kernel='pw_nn_kernel<64,4,0,1> (1x1 conv on float32 MFMA + BN statistics epilogue; the stage-2 / first-unit form)', shape=f'M={M} K=N={Cc}', us=round(t * 1e6, 1),
                    algorithmic_bytes=by, achieved_GBs=round(by / t / 1e9, 1), frac=round(by / t / 1e9 / HBM_PEAK_GBS, 4),
                    cache_state=cold))
    # the form the step runs for the stage-0 / stage-1 forward convs since round 3: the same conv on the bf16 matrix pipe by exact
    # three-way operand splitting (gemm_pw_x3.hip), BatchNorm-apply prologue + statistics epilogue
    stx = torch.rand(4 * G * Cc, device=dev) + 0.5
    wpf = torch.zeros(int(lib.cdrl_pwconv_x3_packed_bytes(Cc)), dtype=torch.uint8, device=dev)
    lib.cdrl_pwconv_x3_pack(P(w), Cc, Cc, Cc, 1, P(wpf), S())
    nbx = int(lib.cdrl_pwconv_x3_partial_rows(G, Mg, Cc, Cc))
    partx = torch.zeros(G * nbx * 2 * Cc, dtype=torch.float64, device=dev)
    t = timeit(lambda k: lib.cdrl_pwconv_x3(P(a[k]), Cc, 0, P(stx), P(wpf), P(bias), P(y[k]), Cc, 0, G, Mg, Cc, Cc, P(partx), S()))
    out.append(dict(kernel='pw_x3_kernel<128,4,BN-apply prologue,statistics epilogue> (1x1 conv, three-way bf16 operand split, v_mfma_f32_32x32x16_bf16)',
                    shape=f'M={M} K=N={Cc}', us=round(t * 1e6, 1), algorithmic_bytes=by, achieved_GBs=round(by / t / 1e9, 1),
                    frac=round(by / t / 1e9 / HBM_PEAK_GBS, 4), cache_state=cold))
    # bf16 path (configuration 3), kernel level: the same conv with bf16 activations + bf16 MFMA (half the bytes)
    ab = [x.to(torch.bfloat16) for x in a]
    yb = [torch.empty(M, Cc, dtype=torch.bfloat16, device=dev) for _ in range(nsets)]
    nbb = int(lib.cdrl_pwconv_bf16_partial_rows(G, Mg, Cc, Cc))
    partb = torch.zeros(G * nbb * 2 * Cc, dtype=torch.float64, device=dev)
    wp = torch.zeros(int(lib.cdrl_pwconv_bf16_packed_elems(Cc)), dtype=torch.bfloat16, device=dev)
    lib.cdrl_pwconv_bf16_pack(P(w), Cc, Cc, P(wp), S())
    t = timeit(lambda k: lib.cdrl_pwconv_bf16(P(ab[k]), Cc, 0, None, None, P(wp), P(bias), P(yb[k]), Cc, 0, G, Mg, Cc, Cc, P(partb), S()))
    byb = 2.0 * M * 2 * Cc
    out.append(dict(kernel='pw_bf16_kernel<128,4,false,true> (bf16 activations, v_mfma_f32_32x32x16_bf16, BN statistics epilogue)',
                    shape=f'M={M} K=N={Cc}', us=round(t * 1e6, 1), algorithmic_bytes=byb, achieved_GBs=round(byb / t / 1e9, 1),
                    frac=round(byb / t / 1e9 / HBM_PEAK_GBS, 4), dtype='bf16',
                    cache_state=f'cold: {nsets} buffer sets ({nsets * byb / 1e6:.0f} MB, above the 256 MB Infinity Cache only from '
                                f'B = 1024; at this size partly cache-resident)'))
    del ab, yb
    dw = torch.empty(Cc, Cc, device=dev)
    ws = torch.empty(int(lib.cdrl_gemm_tn_workspace_elems(M, Cc, Cc)), device=dev)
    t = timeit(lambda k: lib.cdrl_gemm_tn(P(a[k]), Cc, 0, P(y[k]), Cc, 0, P(dw), M, Cc, Cc, P(ws), 0, S()))
    out.append(dict(kernel='tn_lds_kernel<3> + tn_reduce (filter gradient, float32 tensors on the bf16 pipe by three-way split; off the path of '
                           'stages 0 / 1 since the fused conv backward)', shape=f'M={M} K=N={Cc}', us=round(t * 1e6, 1),
                    algorithmic_bytes=by, achieved_GBs=round(by / t / 1e9, 1), frac=round(by / t / 1e9 / HBM_PEAK_GBS, 4),
                    cache_state=cold))
    # stage 2 (K = N = 232, 3x4 pixels: M = B*T*12 rows) -- 6 % of the bytes, 19 % of the step: the forward conv with
    # the BatchNorm statistics epilogue and the filter gradient as the step runs them.  23 MB per launch: these launches are cache-resident
    # at any number of rotating sets the Infinity Cache can be beaten with only from B = 1024 on (labelled)
    C2, M2 = 232, G * B * 12
    a2 = [torch.randn(M2, C2, device=dev) for _ in range(nsets)]
    y2 = [torch.empty(M2, C2, device=dev) for _ in range(nsets)]
    w2 = torch.randn(C2, C2, device=dev)
    b2 = torch.randn(C2, device=dev)
    nb2 = int(lib.cdrl_pwconv_fused_partial_rows(G, M2 // G, C2, C2))
    part2 = torch.zeros(G * nb2 * 2 * C2, dtype=torch.float64, device=dev)
    warm = f'{nsets} rotating buffer sets of {2 * M2 * C2 * 4 / 1e6:.0f} MB = {nsets * 2 * M2 * C2 * 4 / 1e6:.0f} MB: inside the 256 MB Infinity Cache'
    t = timeit(lambda k: lib.cdrl_pwconv_fused(P(a2[k]), C2, 0, None, P(w2), C2, 1, P(b2), P(y2[k]), C2, 0, 0, G, M2 // G, C2, C2, 1, None,
                                               None, P(part2), S()))
    by2 = 4.0 * M2 * 2 * C2
    out.append(dict(kernel='pw_nn_kernel<128,4,0,1> (stage-2 forward conv, K = N = 232, W in registers, float32 MFMA, statistics epilogue)',
                    shape=f'M={M2} K=N={C2}', us=round(t * 1e6, 1), algorithmic_bytes=by2, achieved_GBs=round(by2 / t / 1e9, 1),
                    frac=round(by2 / t / 1e9 / HBM_PEAK_GBS, 4), cache_state=warm))
    # round 6: the form the step runs for these convs now -- one 32-row tile and one block of 128 columns per workgroup, three bf16 planes
    # per operand (pw_x3_wide_kernel), with and without the BatchNorm-apply prologue
    wpx2 = torch.zeros(int(lib.cdrl_pwconv_x3_packed_bytes_n(C2, C2)), dtype=torch.uint8, device=dev)
    lib.cdrl_pwconv_x3_pack(P(w2), C2, C2, C2, 1, P(wpx2), S())
    nbx2 = int(lib.cdrl_pwconv_x3_partial_rows(G, M2 // G, C2, C2))
    partx2 = torch.zeros(G * nbx2 * 2 * C2, dtype=torch.float64, device=dev)
    stx2 = torch.rand(4 * G * C2, device=dev) + 0.5
    for pro, label in ((None, 'statistics epilogue'), (stx2, 'BN-apply prologue, statistics epilogue')):
        t = timeit(lambda k: lib.cdrl_pwconv_x3(P(a2[k]), C2, 0, P(pro), P(wpx2), P(b2), P(y2[k]), C2, 0, G, M2 // G, C2, C2, P(partx2), S()))
        out.append(dict(kernel=f'pw_x3_wide_kernel<{label}> (stage-2 forward conv, K = N = 232, one tile per workgroup, three-way bf16 split)',
                        shape=f'M={M2} K=N={C2}', us=round(t * 1e6, 1), algorithmic_bytes=by2, achieved_GBs=round(by2 / t / 1e9, 1),
                        frac=round(by2 / t / 1e9 / HBM_PEAK_GBS, 4), cache_state=warm))
    dw2 = torch.empty(C2, C2, device=dev)
    ws2 = torch.empty(int(lib.cdrl_gemm_tn_workspace_elems(M2, C2, C2)), device=dev)
    t = timeit(lambda k: lib.cdrl_gemm_tn(P(a2[k]), C2, 0, P(y2[k]), C2, 0, P(dw2), M2, C2, C2, P(ws2), 0, S()))
    out.append(dict(kernel='tn_lds_kernel<3> + tn_reduce (stage-2 filter gradient, K = N = 232)', shape=f'M={M2} K=N={C2}', us=round(t * 1e6, 1),
                    algorithmic_bytes=by2, achieved_GBs=round(by2 / t / 1e9, 1), frac=round(by2 / t / 1e9 / HBM_PEAK_GBS, 4), cache_state=warm))
    del a2, y2
    # round 4: the whole backward of a unit conv in one pass (backward-data + filter / bias gradient + the BatchNorm-backward sums of
    # the BatchNorm in front): reads dz (gathered through the channel shuffle), y, a and writes da
    dz = [torch.randn(M, 2 * Cc, device=dev) for _ in range(nsets)]
    da = [torch.empty(M, Cc, device=dev) for _ in range(nsets)]
    stb = torch.rand(4 * G * Cc, device=dev) + 0.5
    cfb = torch.rand(3 * G * Cc, device=dev) * 0.1
    ga, ba = torch.rand(Cc, device=dev) + 0.5, torch.rand(Cc, device=dev)
    wpx = torch.zeros(int(lib.cdrl_pwconv_x3_packed_bytes(Cc)), dtype=torch.uint8, device=dev)
    lib.cdrl_pwconv_x3_pack(P(w), Cc, Cc, 1, Cc, P(wpx), S())
    qpart = torch.zeros(int(lib.cdrl_pwconv_bwd_fused_workspace(G, Mg, Cc, Cc, 0)), device=dev)
    dbpart = torch.zeros(int(lib.cdrl_pwconv_bwd_fused_workspace(G, Mg, Cc, Cc, 1)), dtype=torch.float64, device=dev)
    gW, gb = torch.empty(Cc, Cc, device=dev), torch.empty(Cc, device=dev)
    adg, adb, acf = torch.empty(Cc, device=dev), torch.empty(Cc, device=dev), torch.empty(3 * G * Cc, device=dev)
    t = timeit(lambda k: lib.cdrl_pwconv_bwd_fused(P(dz[k]), 2 * Cc, Cc, 2 * Cc, 1, P(y[k]), P(stb), P(cfb), P(a[k]), Cc, 0, P(stb), P(ga), P(ba),
                                                   P(adg), P(adb), P(acf), P(w), P(wpx), P(da[k]), Cc, 0, 0, P(gW), P(gb), P(qpart), P(dbpart),
                                                   G, Mg, Cc, Cc, S()))
    byb4 = 4.0 * M * 4 * Cc
    out.append(dict(kernel='pwb_kernel<128,128,shuffle,BN-input> + pwb_reduce_kernel (fused conv backward: da, dW, db, BN2 sums)',
                    shape=f'M={M} K=N={Cc}', us=round(t * 1e6, 1), algorithmic_bytes=byb4, achieved_GBs=round(byb4 / t / 1e9, 1),
                    frac=round(byb4 / t / 1e9 / HBM_PEAK_GBS, 4),
                    cache_state=f'cold: launches rotate over {nsets} buffer sets ({nsets * 5 * M * Cc * 4 / 1e6:.0f} MB > 256 MB Infinity Cache)'))
    del dz, da
    N, Hh, Ww = G * B, 6, 8
    wd = torch.randn(3, 3, Cc, 1, device=dev)
    st = torch.rand(4 * G * Cc, device=dev) + 0.5
    pst = torch.zeros(4 * G * Cc, device=dev)
    ones, zeros = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    mm, mv = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
    wsd = torch.zeros(int(lib.cdrl_dwconv_bn_workspace_doubles(G, B, Hh, Ww, Cc, 1)), dtype=torch.float64, device=dev)
    t = timeit(lambda k: lib.cdrl_dwconv_bn_fwd(P(a[k]), P(st), P(wd), P(bias), P(y[k]), G, B, Hh, Ww, Cc, 1, P(ones), P(zeros), P(mm),
                                                P(mv), 1, P(pst), P(wsd), S()))
    byd = 4.0 * 2 * N * Hh * Ww * Cc
    out.append(dict(kernel='dwf_fwd_kernel<1,4,true> + bn_finalize (BN+ReLU6 -> dw3x3 -> BN statistics)', shape=f'{N}x{Hh}x{Ww}x{Cc}',
                    us=round(t * 1e6, 1), algorithmic_bytes=byd, achieved_GBs=round(byd / t / 1e9, 1),
                    frac=round(byd / t / 1e9 / HBM_PEAK_GBS, 4), cache_state=cold))
    # round 5: the stem conv + BatchNorm statistics of the benchmark shape (stem_fwd_band_kernel: image band staged in LDS); one buffer set
    # is 399 MB (images 133 MB read, conv output 265 MB written) > the 256 MB Infinity Cache, so a single set is cold already
    Hs, Ws, Cs = 90, 120, 24
    Ho, Wo = (Hs - 3) // 2 + 1, (Ws - 3) // 2 + 1
    xs = torch.rand(B, T, Hs, Ws, 3, device=dev)
    ws_ = torch.randn(3, 3, 3, Cs, device=dev) * 0.3
    bs_ = torch.randn(Cs, device=dev)
    ys = torch.empty(B * T, Ho, Wo, Cs, device=dev)
    rows = int(lib.cdrl_stem_fwd_stats_rows(B, T, Hs, Ws, Cs))
    parts = torch.zeros(T * rows * 2 * Cs, dtype=torch.float64, device=dev)
    t = timeit(lambda k: lib.cdrl_stem_fwd_stats(P(xs), P(ws_), P(bs_), P(ys), P(parts), B, T, Hs, Ws, Cs, S()), iters=16)
    bys = 4.0 * (xs.numel() + ys.numel())
    out.append(dict(kernel='stem_fwd_band_kernel<2,6> (3x3/s2 stem conv, 3 -> 24 channels, + BatchNorm statistics; image band staged in LDS)',
                    shape=f'{B}x{T}x{Hs}x{Ws}x3 -> {B * T}x{Ho}x{Wo}x{Cs}', us=round(t * 1e6, 1), algorithmic_bytes=bys,
                    achieved_GBs=round(bys / t / 1e9, 1), frac=round(bys / t / 1e9 / HBM_PEAK_GBS, 4),
                    cache_state='cold: one buffer set (399 MB) exceeds the 256 MB Infinity Cache'))
    return out


def secondary_config3(dev, steps=24, warmup=4):
    """BASELINE.json configs[2] measured in the SAME process right after the headline timing, so that the driver's single
    invocation carries a driver-timed figure for it: the same update-step at B = 1024 with bf16 activation storage + bf16 MFMA
    operands in the image tower (LearnerEngine(compute='bf16s')), inputs resident in HBM, HIP-event timed on the launch stream.
    Roofline against the bf16 algorithmic bytes (48.14 GB per update-step), PMC traffic from the committed run of that workload."""
    import torch
    from carla_driving_rl_agent_amd import synthetic
    from carla_driving_rl_agent_amd.engine import LearnerEngine, gae_returns
    from carla_driving_rl_agent_amd.init import init_engine_parameters
    B, T, H, W = 1024, 4, 90, 120
    eng = LearnerEngine(B, device=dev, T=T, H=H, W=W, compute='bf16s')
    init_engine_parameters(eng, seed=42)
    r = synthetic.make_rollout(B, T=T, H=H, W=W, seed=43)
    states = {k: torch.as_tensor(v).to(dev) for k, v in r['states'].items()}
    rewards = torch.cat([torch.as_tensor(r['reward']).to(dev), torch.zeros(1, device=dev)])
    values = torch.cat([torch.as_tensor(r['value']).to(dev), torch.zeros((1, 2), device=dev)])
    _, returns_be, _, adv = gae_returns(rewards, values, synthetic.DEFAULT_HP['gamma'], synthetic.DEFAULT_HP['lambda_'],
                                        synthetic.DEFAULT_HP['advantage_scale'])
    speed = (torch.as_tensor(r['speed'][:, 0]) / 100.0).to(dev).contiguous()
    sim = torch.as_tensor(r['similarity'][:, 0]).to(dev).contiguous()
    pol = dict(states=states, advantages=adv.contiguous(), old_log_prob=torch.as_tensor(r['old_log_prob']).to(dev), speed=speed,
               similarity=sim, u=torch.as_tensor(r['action']).to(dev), du_da=None, du_db=None)
    val = dict(states=states, returns=returns_be.contiguous(), speed=speed, similarity=sim)

    def one(k):
        eng.policy_forward_backward_resample(pol, seed=42, offset=k)
        eng.policy_apply()
        eng.value_forward_backward(val)
        eng.value_apply()

    for k in range(warmup):
        one(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time()
    e0.record()
    for k in range(steps):
        one(warmup + k)
    e1.record()
    torch.cuda.synchronize()
    wall = (time.time() - t0) / steps
    dev_s = e0.elapsed_time(e1) * 1e-3 / steps
    alg = alg_bytes_per_update_step(B, T, H, W, 2)
    loss = eng.metrics('policy')['loss']
    del eng
    torch.cuda.empty_cache()
    return dict(config='configs[2]: same network, bf16 activation storage + bf16 MFMA operands in the image tower, batch 1024, 1 GPU',
                per_gpu_batch=B, dtype='bf16 (activation storage + MFMA operands of the image tower; f32 accumulate / statistics / weights)',
                steps=steps, warmup=warmup, ms_per_step=round(wall * 1e3, 3), device_ms_per_step=round(dev_s * 1e3, 3),
                update_steps_per_s=round(1.0 / wall, 3), equivalent_256_sample_update_steps_per_s=round(B / 256.0 / wall, 2),
                roofline=dict(bound='hbm', achieved=round(alg / dev_s / 1e9, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                              frac=round(alg / dev_s / 1e9 / HBM_PEAK_GBS, 4), traffic=pmc_traffic(B, T, H, W, 'bf16s'),
                              algorithmic_bytes_per_launch=alg), final_policy_loss=loss)


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown CPU'


def _oracle_timer(B_sample, T, H, W, threads, warmup, steps, budget_s, seed=42):
    import torch
    from oracle import model as OM
    from oracle.spec import NetConfig, trunk_spec, policy_spec, value_spec
    from carla_driving_rl_agent_amd import synthetic
    from tests.util import make_batches, oracle_batch
    torch.set_num_threads(threads)
    cfg = NetConfig(T=T, H=H, W=W)
    learner = OM.OracleLearner(cfg, OM.init_params(trunk_spec(cfg), 1, False), OM.init_params(policy_spec(cfg), 2, False),
                               OM.init_params(value_spec(cfg), 3, False), synthetic.DEFAULT_HP)
    pol, val = make_batches(B_sample, H, W, seed=seed, faithful=True)       # injected sample + Jacobians = the re-sampled loss
    pol, val = oracle_batch(pol), oracle_batch(val)
    t_start = time.time()
    for _ in range(warmup):
        learner.policy_step(pol)
        learner.value_step(val)
    times = []
    while len(times) < steps and (not times or (time.time() - t_start) < budget_s):
        t0 = time.time()
        learner.policy_step(pol)
        learner.value_step(val)
        times.append(time.time() - t0)
    return times


def cpu_baseline(B_sample, T, H, W, threads, one_thread_batch=32):
    """Oracle (PyTorch-CPU restatement of the reference TF-CPU path: same per-slice BatchNorm, same two-phase update) timed
    on the host cores, SURVEY.md section 8(d): the benchmark's own minibatch (B_sample = 256 by default), 2 warm-ups, then
    >= 5 timed update-steps (bounded by a wall-clock budget), median and min, at the best thread count; plus a bounded
    1-thread figure (one update-step at a smaller minibatch, scaled)."""
    import statistics
    times = _oracle_timer(B_sample, T, H, W, threads, warmup=2, steps=5, budget_s=150.0)
    med, best = statistics.median(times), min(times)
    one = _oracle_timer(one_thread_batch, T, H, W, 1, warmup=0, steps=1, budget_s=1.0)[0]
    scale = B_sample / 256.0
    logical = os.cpu_count() or 0
    try:        # physical cores = distinct (package, core id) pairs
        phys, pkg = set(), None
        for line in open('/proc/cpuinfo'):
            if line.startswith('physical id'):
                pkg = line.split(':')[1].strip()
            elif line.startswith('core id'):
                phys.add((pkg, line.split(':')[1].strip()))
        physical = len(phys) or None
    except Exception:
        physical = None
    return dict(value=scale / med, unit='update-steps/s (256-sample update-steps)', cores=threads, kind='port',
                host_logical_cores=logical, host_physical_cores=physical,
                median_s_per_step=round(med, 3), min_s_per_step=round(best, 3), timed_steps=len(times), warmup_steps=2,
                one_thread=dict(value=(one_thread_batch / 256.0) / one, s_per_step=round(one, 3), minibatch=one_thread_batch,
                                note=f'1 update-step at minibatch {one_thread_batch} on 1 thread, scaled by {one_thread_batch}/256'),
                sample=f'{len(times)} timed update-steps (after 2 warm-ups) of the PyTorch-CPU oracle at minibatch {B_sample} '
                       f'(T={T}, {H}x{W}x3, re-sampled policy loss with injected sample), {threads} intra-op threads '
                       f'(fastest measured setting on this host class), median {med:.3f} s, min {best:.3f} s per update-step; '
                       f'host: {_cpu_model()}, {os.cpu_count()} logical cores; TensorFlow itself is not installable here')


def run_cpu_baseline_child(B_sample, T, H, W, threads, timeout=420):
    """The oracle runs in a CHILD process (plain subprocess, never exec from the GPU-initialised
    process) with a hard wall-clock limit, so the default bench always finishes within minutes."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES='')
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-only', '--cpu-sample-batch', str(B_sample),
           '--cpu-threads', str(threads), '--height', str(H), '--width', str(W)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith('{'):
                return json.loads(line)
        return dict(value=None, unit='update-steps/s', cores=threads, kind='port', sample=f'failed rc={r.returncode}: {r.stderr[-300:]}')
    except subprocess.TimeoutExpired:
        return dict(value=None, unit='update-steps/s', cores=threads, kind='port', sample=f'timed out after {timeout}s')


class cpu_rollout_rows:
    """CPU stand-ins of the rollout-side rows for tools/bench_rollout_rows.py: the oracle -- the
    reference's algorithm restated in PyTorch / numpy -- on the host.  Part of the benchmark's baseline leg (like cpu_baseline
    above), never of the product path."""

    def __init__(self, H=90, W=120, A=2, threads=16):
        import torch as _t
        from oracle import model as OM
        from oracle.spec import NetConfig, trunk_spec, policy_spec, value_spec
        from carla_driving_rl_agent_amd import synthetic
        _t.set_num_threads(threads)
        cfg = NetConfig(H=H, W=W, A=A)
        self._oracle = OM.OracleLearner(cfg, OM.init_params(trunk_spec(cfg), 1), OM.init_params(policy_spec(cfg), 2),
                                        OM.init_params(value_spec(cfg), 3), dict(synthetic.DEFAULT_HP))
        self._synthetic = synthetic

    def predict(self, states):
        return self._oracle.predict(states)

    def beta(self, alpha, beta):
        import numpy as _np
        return self._synthetic.beta_sample_with_jacobian(alpha, beta, _np.random.default_rng(0))

    def augment(self, stack, plan):
        from oracle import augment as OA
        return OA.augment(stack, plan)

    def gae(self, rewards, values_be, gamma, lambda_):
        from oracle import gae as OG
        return OG.compute_returns(rewards, gamma), OG.compute_advantages(rewards, values_be, gamma, lambda_, 2.0)


def spawn_ranks(n, argv):
    """`bench.py --gpus N` started as a plain process: start the N ranks as CHILDREN through torch.distributed.run (one
    process per GPU, RCCL rendezvous on 127.0.0.1) BEFORE anything in this process touches the GPU, relay their output and
    exit with their status.  (The driver's own `python -m torch.distributed.run ... bench.py --gpus N` form sets
    WORLD_SIZE and never comes through here.)"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200,
                    help='timed update-steps (default 200: a timed region of ~3 s, long enough for an external GPU-busy sampler)')
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--height', type=int, default=90)
    ap.add_argument('--width', type=int, default=120)
    ap.add_argument('--dtype', choices=['f32', 'bf16', 'bf16s'], default='f32',
                    help="bf16s: configuration 3 (bf16 activation STORAGE in the image tower + bf16 MFMA operands in its 1x1 convolutions, "
                         "float32 accumulation / statistics / weights); bf16: the operand mode alone (float32 tensors); quote either with "
                         "--batch 1024")
    ap.add_argument('--rollout-rows', action='store_true',
                    help='measure the rollout-side rows (predict for E environments, Beta sampling, augmentation, GAE, checkpoint I/O, '
                         'the agent-level collect / update cycle) with their CPU stand-ins and print that JSON instead of the benchmark line')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-rooflines', action='store_true',
                    help='skip the isolated-kernel roofline launches (profiling passes: PMC totals and trace call counts then contain '
                         'only the update-steps)')
    ap.add_argument('--no-secondary', action='store_true',
                    help='skip the configs[2] measurement (bf16 storage, B = 1024) that the default run appends as `secondary`')
    ap.add_argument('--cpu-sample-batch', type=int, default=256)
    ap.add_argument('--cpu-threads', type=int, default=min(16, os.cpu_count() or 1),
                    help='oracle intra-op threads; measured on the 2x64-core EPYC GPU host: 16 threads is the '
                         'fastest setting (32: 2.2x slower, 64: 5.8x slower, 256: does not finish)')
    ap.add_argument('--cpu-baseline-only', action='store_true')
    ap.add_argument('--stored-actions', action='store_true',
                    help='time the textbook-PPO loss on the stored rollout actions instead of the reference-faithful re-sampled loss')
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args.cpu_sample_batch, 4, args.height, args.width, args.cpu_threads)))
        return
    if args.rollout_rows:
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import bench_rollout_rows
        bench_rollout_rows.main(cpu=cpu_rollout_rows())
        return
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    from carla_driving_rl_agent_amd import synthetic
    from carla_driving_rl_agent_amd.engine import LearnerEngine, gae_returns
    from carla_driving_rl_agent_amd.parallel import DataParallelLearner

    from carla_driving_rl_agent_amd import _lib as _cdrl_lib
    # diagnostic switches that skip work or synchronisation (CDRL_DIAG_*, honoured only with CDRL_DIAG=1) give wrong results:
    # a benchmark line measured under one is refused, and every CDRL_* override in effect is written into the line
    if _cdrl_lib.diag_active():
        raise SystemExit(f'[bench] refusing to benchmark: wrong-result diagnostic switches are active ({_cdrl_lib.env_overrides()})')
    env_overrides = _cdrl_lib.env_overrides()
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit(f'[bench] WORLD_SIZE={world} but --gpus {args.gpus}: start one rank per GPU')
    # Functional check of the N > 1 path on a ONE-GPU box (tests/test_gpu_dp_rccl.py): CDRL_BENCH_SHARE_DEVICE=1 puts every rank on
    # device 0 and routes the collectives of the device tensors through gloo (RCCL refuses two ranks on one device).  The line it
    # prints is marked `shared_device` and is NOT a scaling figure.
    share_device = world > 1 and os.environ.get('CDRL_BENCH_SHARE_DEVICE') == '1'
    if share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = f'cuda:{local_rank}'
    use_dist = world > 1 or ('RANK' in os.environ and os.environ.get('CDRL_FORCE_COLLECTIVES') == '1')
    if use_dist:      # launched by torchrun: RCCL over xGMI (CDRL_FORCE_COLLECTIVES=1 exercises it on 1 GPU too)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if share_device:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device(dev))

    B, T, H, W = args.batch, 4, args.height, args.width
    eng = LearnerEngine(B, device=dev, T=T, H=H, W=W, compute=args.dtype)
    # random-init weights of the reference architecture, identical on every rank
    from carla_driving_rl_agent_amd.init import init_engine_parameters
    init_engine_parameters(eng, seed=42)
    dp = DataParallelLearner(eng, force_collectives=use_dist and world == 1)
    dp.broadcast_parameters()

    # rollout shard of this rank (weak scaling: B timesteps per GPU), already resident in HBM
    n = B
    r = synthetic.make_rollout(n, T=T, H=H, W=W, seed=42 + rank)
    states = {k: torch.as_tensor(v).to(dev) for k, v in r['states'].items()}
    rewards = torch.as_tensor(r['reward']).to(dev)
    values = torch.as_tensor(r['value']).to(dev)
    # end_trajectory: bootstrap with the terminal last_value (0, 0)  (reference core/networks.py:171,214-216)
    rewards = torch.cat([rewards, torch.zeros(1, device=dev)])
    values = torch.cat([values, torch.zeros((1, 2), device=dev)])
    torch.cuda.synchronize()
    t0 = time.time()
    returns, returns_be, adv_raw, adv = gae_returns(rewards, values, synthetic.DEFAULT_HP['gamma'],
                                                    synthetic.DEFAULT_HP['lambda_'], synthetic.DEFAULT_HP['advantage_scale'])
    torch.cuda.synchronize()
    gae_ms = (time.time() - t0) * 1e3
    speed = (torch.as_tensor(r['speed'][:, 0]) / 100.0).to(dev).contiguous()
    sim = torch.as_tensor(r['similarity'][:, 0]).to(dev).contiguous()
    pol = dict(states=states, advantages=adv.contiguous(), old_log_prob=torch.as_tensor(r['old_log_prob']).to(dev),
               speed=speed, similarity=sim, u=torch.as_tensor(r['action']).to(dev), du_da=None, du_db=None)
    val = dict(states=states, returns=returns_be.contiguous(), speed=speed, similarity=sim)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # the policy loss is the reference's: evaluated on a fresh Beta sample of the NEW policy drawn on the device with pathwise
    # gradients (core/networks.py:96-110); Philox stream = (seed, rank-disjoint offset per step)
    step_no = [0]

    def one_step():
        step_no[0] += 1
        dp.update_step(pol, val, resample=None if args.stored_actions else (42, step_no[0] * world + rank))

    for _ in range(args.warmup):
        one_step()
    barrier()
    # EXACTLY args.steps timed update-steps between two barrier + synchronize brackets; HIP events on the launch stream split
    # them into (up to) 5 blocks so that the line can carry min / median per-step times besides the mean
    nblk = min(5, args.steps)
    edges = [round(i * args.steps / nblk) for i in range(nblk + 1)]
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(nblk + 1)]
    t0 = time.time()
    evs[0].record()
    host_n, host_t = min(6, args.steps), None
    done = 0
    for b in range(nblk):
        for _ in range(edges[b + 1] - edges[b]):
            one_step()
            done += 1
            if done == host_n:
                # host time to ENQUEUE one update-step, over the first few steps of the timed region only: the launch queues are
                # empty then -- later the host runs ahead until the queues are full and is throttled to the device's pace
                host_t = (time.time() - t0) / host_n
        evs[b + 1].record()
    host_ms = host_t * 1e3
    barrier()
    elapsed = time.time() - t0
    dev_ms = evs[0].elapsed_time(evs[-1])
    block_ms = [evs[b].elapsed_time(evs[b + 1]) / (edges[b + 1] - edges[b]) for b in range(nblk)]
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_p = eng.metrics('policy')['loss']
    loss_v = eng.metrics('value')['loss']

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * args.steps / elapsed
        alg = alg_bytes_per_update_step(B, T, H, W, 2 if args.dtype == 'bf16s' else 4) if (H, W) in ALG_ELEMS_PER_FRAME else None
        dev_s_per_step = dev_ms * 1e-3 / args.steps
        roof = None
        if alg is not None:
            achieved = alg / dev_s_per_step / 1e9
            roof = dict(bound='hbm', achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                        frac=round(achieved / HBM_PEAK_GBS, 4), traffic=pmc_traffic(B, T, H, W, args.dtype),
                        kernel='one PPO update-step (all launches of the step, HIP-event timed on the launch stream)',
                        algorithmic_bytes_per_launch=alg)
        out = dict(metric='PPO update-steps/sec (batch=256, 4x90x120x3 obs)', value=round(value, 3), unit='update-steps/s',
                   n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_per_step, 3),
                   higher_is_better=True, scaling='weak', vs_baseline=None,
                   dtype={'f32': 'f32', 'bf16': 'bf16 MFMA operands in the 1x1 convolutions (fwd, bwd-data, filter gradient), f32 storage/accumulate',
                          'bf16s': 'bf16 (activation storage + MFMA operands of the image tower; f32 accumulate / statistics / weights)'}[args.dtype],
                   data='synthetic',
                   config=dict(workload=f'configs[1]: synthetic rollout buffer {B}x{T}-frame {H}x{W}x3 obs per GPU, full '
                                        f'CARLANetwork fwd/bwd + PPO (re-sampled Beta, pathwise) / value loss + clip + Adam, ' +
                                        {'f32': 'fp32', 'bf16': 'bf16-operand compute mode (configs[2] arithmetic, float32 tensors)',
                                         'bf16s': 'bf16 storage + bf16 MFMA (configs[2])'}[args.dtype],
                               per_gpu_batch=B, global_batch=B * world, time_horizon=T, image=[H, W, 3],
                               parallelism=f'dp{world}', passes_per_step=2,
                               policy_loss='stored-actions' if args.stored_actions else 'resampled (reference-faithful)'),
                   roofline=roof, mfma=pmc_mfma(ms_per_step), dominant_kernel=dominant_kernel(B, T, H, W), kernel_rooflines=kernel_rooflines(B, T) if (world == 1 and not args.no_kernel_rooflines) else None, gae_ms=round(gae_ms, 3), device_ms_per_step=round(dev_ms / args.steps, 3),
                   device_ms_per_step_blocks=dict(blocks=[round(x, 3) for x in block_ms], min=round(min(block_ms), 3), median=round(sorted(block_ms)[len(block_ms) // 2], 3)),
                   host_enqueue_ms_per_step=round(host_ms, 3), env_overrides=env_overrides,
                   **(dict(shared_device='all ranks on cuda:0, collectives through gloo: a functional check of the N > 1 path, not a scaling figure') if share_device else {}),
                   final_losses=dict(policy=loss_p, value=loss_v))
        # (--no-kernel-rooflines marks a profiling pass: PMC totals and trace call counts must contain only the headline update-steps)
        if world == 1 and not args.no_secondary and not args.no_kernel_rooflines and (B, H, W, args.dtype) == (256, 90, 120, 'f32'):
            del dp, eng, states, pol, val                   # (the headline engine's 6 GB workspace is not needed any more)
            torch.cuda.empty_cache()
            try:
                out['secondary'] = secondary_config3(dev)
            except Exception as e:                          # the headline line must survive a failure of the extra measurement
                out['secondary'] = dict(error=repr(e))
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = run_cpu_baseline_child(args.cpu_sample_batch, T, H, W, args.cpu_threads)
    else:
        out = None
    # RCCL writes a version banner through C stdio (block-buffered when stdout is a pipe): every rank flushes it BEFORE
    # the final barrier so that rank 0's JSON line is the last line on the job's stdout
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
