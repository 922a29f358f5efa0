"""GPU tool: the hand-written kernels alone, at the shapes the batch-8 512x512 step gives them.

    python tools/kernel_microbench.py time [reps]     HIP-event timing per C-ABI call  -> one JSON line per case
    python tools/kernel_microbench.py pmc             2 calls per case, each case preceded by a MARKER launch

The pmc mode is what rocprofv3 --pmc passes are run over (tools/gpu_session.sh); tools/summarise_pmc.py maps every
dispatch of the counter CSVs back to its case through the markers (a skd_leaky_relu launch whose grid size encodes
the case id), so per-case HBM bytes / MFMA counters need no guessing from grid sizes.

Cases = the channels-last TRAINING InPlace-ABN entries on the student's layers, the channels-last inference apply on
the teacher's layers, the pair-wise Gram / backward kernels at M in {9, 1089, 4225}, and (round 4) one isolated row per
remaining kernel family: spectral norm (single layer and all four layers of D per launch), pixel-wise KL, fused
upsample + CE, channels-last pyramid pooling / concat, the stem max-pool, the pair-wise pooling, the evaluation tail.  Each case states its
ALGORITHMIC bytes or flops per call; the manifest is printed as the first JSON line.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timing import warm_timed  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

STUDENT = [(8 * 256 * 256, 64), (8 * 256 * 256, 128), (8 * 129 * 129, 64), (8 * 65 * 65, 128), (8 * 65 * 65, 256),
           (8 * 65 * 65, 512)]
TEACHER = [(8 * 65 * 65, 1024), (8 * 65 * 65, 2048), (8 * 65 * 65, 256), (8 * 129 * 129, 256), (8 * 65 * 65, 512)]
PAIRWISE_M = [9, 1089, 4225]


def build_cases(lib, torch, dev, st):
    p = lambda t: None if t is None else t.data_ptr()
    cases = []

    def add(name, shape, fn, bytes_=None, flops=None, keep=()):
        cases.append({"id": len(cases), "name": name, "shape": shape, "algo_bytes": bytes_, "algo_flops": flops,
                      "fn": fn, "keep": keep})

    for rows, C in STUDENT:
        n = rows * C
        x, r, dz, out, dx, dres = (torch.randn(rows, C, device=dev) for _ in range(6))
        w, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        m, v = torch.empty(C, device=dev), torch.empty(C, device=dev)
        e, ey = torch.empty(C, device=dev), torch.empty(C, device=dev)
        dw, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        ws = torch.empty(lib.skd_abn_nhwc_workspace_floats(rows, C), device=dev)
        keep = (x, r, dz, out, dx, dres, w, b, rm, rv, m, v, e, ey, dw, db, ws)
        sh = [rows, C]
        add("stats_nhwc", sh, lambda x=x, m=m, v=v, ws=ws, rows=rows, C=C: lib.skd_abn_stats_nhwc(rows, C, p(x), p(m), p(v), p(ws), st), 4 * n, keep=keep)
        add("forward_train_nhwc(leaky,in place)", sh, lambda x=x, w=w, b=b, rm=rm, rv=rv, m=m, v=v, ws=ws, rows=rows, C=C:
            lib.skd_abn_forward_train_nhwc(rows, C, p(x), None, p(x), p(w), p(b), p(rm), p(rv), p(m), p(v), 0.1, 1e-5, 1, 0.01, p(ws), st), 12 * n)
        add("forward_train_nhwc(relu,to out)", sh, lambda x=x, out=out, w=w, b=b, rm=rm, rv=rv, m=m, v=v, ws=ws, rows=rows, C=C:
            lib.skd_abn_forward_train_nhwc(rows, C, p(x), None, p(out), p(w), p(b), p(rm), p(rv), p(m), p(v), 0.1, 1e-5, 3, 0.0, p(ws), st), 12 * n)
        add("forward_train_nhwc(relu,+residual)", sh, lambda x=x, r=r, out=out, w=w, b=b, rm=rm, rv=rv, m=m, v=v, ws=ws, rows=rows, C=C:
            lib.skd_abn_forward_train_nhwc(rows, C, p(x), p(r), p(out), p(w), p(b), p(rm), p(rv), p(m), p(v), 0.1, 1e-5, 3, 0.0, p(ws), st), 16 * n)
        add("backward_reduce_nhwc(leaky)", sh, lambda x=x, dz=dz, w=w, b=b, e=e, ey=ey, ws=ws, rows=rows, C=C:
            lib.skd_abn_backward_reduce_nhwc(rows, C, p(x), p(dz), p(w), p(b), p(e), p(ey), 1e-5, 1, 0.01, p(ws), st), 8 * n)
        add("backward_dx_nhwc(leaky)", sh, lambda x=x, dz=dz, v=rv, w=w, b=b, e=e, ey=ey, dx=dx, dw=dw, db=db, rows=rows, C=C:
            lib.skd_abn_backward_dx_nhwc(rows, C, p(x), p(dz), p(v), p(w), p(b), p(e), p(ey), p(dx), p(dw), p(db), 1e-5, 1, 0.01, 0, st), 12 * n)
        add("relu_backward_reduce_nhwc", sh, lambda x=x, out=out, dz=dz, m=rm, v=rv, e=e, ey=ey, ws=ws, rows=rows, C=C:
            lib.skd_abn_relu_backward_reduce_nhwc(rows, C, p(x), p(out), p(dz), p(m), p(v), p(e), p(ey), 1e-5, p(ws), st), 12 * n)
        add("relu_backward_reduce_nhwc_x(mask from x)", sh, lambda x=x, dz=dz, m=rm, v=rv, w=w, b=b, e=e, ey=ey, ws=ws, rows=rows, C=C:
            lib.skd_abn_relu_backward_reduce_nhwc_x(rows, C, p(x), p(dz), p(m), p(v), p(w), p(b), p(e), p(ey), 1e-5, p(ws), st), 8 * n)
        add("relu_backward_dx_nhwc_x(mask from x)", sh, lambda x=x, dz=dz, m=rm, v=rv, w=w, b=b, e=e, ey=ey, dx=dx, dw=dw, db=db, rows=rows, C=C:
            lib.skd_abn_relu_backward_dx_nhwc_x(rows, C, p(x), p(dz), p(m), p(v), p(w), p(b), p(e), p(ey), p(dx), p(dw), p(db), 1e-5, 0, st), 12 * n)
        add("relu_backward_dx_nhwc", sh, lambda x=x, out=out, dz=dz, m=rm, v=rv, w=w, e=e, ey=ey, dx=dx, dw=dw, db=db, rows=rows, C=C:
            lib.skd_abn_relu_backward_dx_nhwc(rows, C, p(x), p(out), p(dz), p(m), p(v), p(w), p(e), p(ey), p(dx), None, p(dw), p(db), 1e-5, 0, st), 16 * n)
        add("relu_backward_dx_nhwc(+dres)", sh, lambda x=x, out=out, dz=dz, m=rm, v=rv, w=w, e=e, ey=ey, dx=dx, dres=dres, dw=dw, db=db, rows=rows, C=C:
            lib.skd_abn_relu_backward_dx_nhwc(rows, C, p(x), p(out), p(dz), p(m), p(v), p(w), p(e), p(ey), p(dx), p(dres), p(dw), p(db), 1e-5, 0, st), 20 * n)
        # reduce + dx in one call (one register-resident launch when the tensor fits): bytes = the two-pass algorithmic figure,
        # so the GB/s are "equivalent" rates comparable with the separate entries above
        add("backward_nhwc(leaky, one call)", sh, lambda x=x, dz=dz, v=rv, w=w, b=b, e=e, ey=ey, dx=dx, dw=dw, db=db, ws=ws, rows=rows, C=C:
            lib.skd_abn_backward_nhwc(rows, C, p(x), p(dz), p(v), p(w), p(b), p(e), p(ey), p(dx), p(dw), p(db), 1e-5, 1, 0.01, 0, p(ws), st), 20 * n)
        add("relu_backward_nhwc(mask from x, one call)", sh, lambda x=x, dz=dz, m=rm, v=rv, w=w, b=b, e=e, ey=ey, dx=dx, dw=dw, db=db, ws=ws, rows=rows, C=C:
            lib.skd_abn_relu_backward_nhwc(rows, C, p(x), None, p(dz), p(m), p(v), p(w), p(b), p(e), p(ey), p(dx), None, p(dw), p(db), 1e-5, 0, p(ws), st), 20 * n)
        add("relu_backward_nhwc(+dres, one call)", sh, lambda x=x, out=out, dz=dz, m=rm, v=rv, w=w, b=b, e=e, ey=ey, dx=dx, dres=dres, dw=dw, db=db, ws=ws, rows=rows, C=C:
            lib.skd_abn_relu_backward_nhwc(rows, C, p(x), p(out), p(dz), p(m), p(v), p(w), p(b), p(e), p(ey), p(dx), p(dres), p(dw), p(db), 1e-5, 0, p(ws), st), 32 * n)
    for rows, C in TEACHER:
        n = rows * C
        x, r = torch.randn(rows, C, device=dev), torch.randn(rows, C, device=dev)
        w, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        keep = (x, r, w, b, rm, rv)
        add("apply_nhwc(eval,relu)", [rows, C], lambda x=x, w=w, b=b, rm=rm, rv=rv, rows=rows, C=C:
            lib.skd_abn_apply_nhwc(rows, C, p(x), None, p(rm), p(rv), p(w), p(b), 1e-5, 3, 0.01, st), 8 * n, keep=keep)
        add("apply_nhwc(eval,relu,+residual)", [rows, C], lambda x=x, r=r, w=w, b=b, rm=rm, rv=rv, rows=rows, C=C:
            lib.skd_abn_apply_nhwc(rows, C, p(x), p(r), p(rm), p(rv), p(w), p(b), 1e-5, 3, 0.01, st), 12 * n)
    B, Cs, Ct = 8, 128, 512
    for M in PAIRWISE_M:
        ldm = lib.skd_pairwise_ldm(M)
        ps, pt = torch.randn(B, Cs, M, device=dev), torch.randn(B, Ct, M, device=dev)
        fs, ft = torch.empty(B, Cs, ldm, device=dev), torch.empty(B, Ct, ldm, device=dev)
        nrm = torch.empty(B, M, device=dev)
        G, loss, gl = torch.empty(B, ldm, ldm, device=dev), torch.empty(1, device=dev), torch.ones(1, device=dev)
        dp = torch.empty(B, Cs, ldm, device=dev)
        ws = torch.empty(max(1, lib.skd_pairwise_workspace_floats(B, M)), device=dev)
        bws = torch.empty(max(1, lib.skd_pairwise_backward_workspace_floats(B, Cs, M)), device=dev)
        keep = (ps, pt, fs, ft, nrm, G, loss, gl, dp, ws, bws)
        lib.skd_channel_l2_normalise(B, Ct, M, p(pt), p(ft), ldm, None, 0, None, st)
        # algorithmic bytes: read the pooled features, write the normalised panel (the node-major copy of rounds 1-2 is gone)
        add("l2_normalise(student)", [B, Cs, M], lambda ps=ps, fs=fs, nrm=nrm, M=M, ldm=ldm:
            lib.skd_channel_l2_normalise(B, Cs, M, p(ps), p(fs), ldm, None, 0, p(nrm), st), 4 * B * Cs * M * 2, keep=keep)
        add("pairwise_gram_loss", [B, Cs, Ct, M], lambda fs=fs, ft=ft, G=G, loss=loss, ws=ws, M=M, ldm=ldm:
            lib.skd_pairwise_gram_loss(B, Cs, Ct, M, ldm, p(fs), p(ft), p(G), p(loss), p(ws), st), flops=2.0 * B * M * M * (Cs + Ct))
        add("pairwise_gram_loss(no G store)", [B, Cs, Ct, M], lambda fs=fs, ft=ft, loss=loss, ws=ws, M=M, ldm=ldm:
            lib.skd_pairwise_gram_loss(B, Cs, Ct, M, ldm, p(fs), p(ft), None, p(loss), p(ws), st), flops=2.0 * B * M * M * (Cs + Ct))
        add("pairwise_backward", [B, Cs, M], lambda fs=fs, G=G, nrm=nrm, gl=gl, dp=dp, bws=bws, M=M, ldm=ldm:
            lib.skd_pairwise_backward(B, Cs, M, ldm, p(fs), p(G), p(nrm), p(gl), p(dp), p(bws), st), flops=2.0 * B * M * M * Cs)
    # ---- the other hand-written kernel families (every .hip file gets an isolated roofline row) ----
    import ctypes
    iarr = lambda v: (ctypes.c_int * len(v))(*v)
    parr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    # spectral norm (csrc/spectral.hip): algorithmic bytes per layer-forward = 4 x numel(W) x 4 (two mat-vec reads, read + write of
    # W / sigma), backward the same (read gW, W twice, write gW_bar)
    SN = [(64, 19 * 16), (128, 64 * 16), (256, 128 * 16), (512, 256 * 16)]
    sn_t = []
    for h, w in SN:
        W, out, gw, gwb = (torch.randn(h, w, device=dev) * 0.05 for _ in range(4))
        u, v, sg = torch.randn(h, device=dev), torch.randn(w, device=dev), torch.ones(1, device=dev)
        ws = torch.empty(max(1, lib.skd_spectral_workspace_floats(h, w)), device=dev)
        sn_t.append((W, out, gw, gwb, u, v, sg, ws))
        add("spectral_norm_forward", [h, w], lambda W=W, out=out, u=u, v=v, sg=sg, ws=ws, h=h, w=w:
            lib.skd_spectral_norm_forward(h, w, p(W), p(u), p(v), p(sg), p(out), p(ws), st), 16 * h * w, keep=sn_t[-1])
        add("spectral_norm_backward", [h, w], lambda W=W, gw=gw, gwb=gwb, u=u, v=v, sg=sg, ws=ws, h=h, w=w:
            lib.skd_spectral_norm_backward(h, w, p(W), p(u), p(v), p(sg), p(gw), p(gwb), p(ws), st), 16 * h * w)
    hs, ws_ = iarr([h for h, _ in SN]), iarr([w for _, w in SN])
    work = torch.empty(sum(max(1, lib.skd_spectral_workspace_floats(h, w)) for h, w in SN), device=dev)
    sig = torch.ones(len(SN), device=dev)
    sigl = [sig[k:k + 1] for k in range(len(SN))]
    tot = sum(h * w for h, w in SN)
    add("spectral_norm_forward_multi(4 layers of D)", [len(SN), tot], lambda:
        lib.skd_spectral_norm_forward_multi(len(SN), hs, ws_, parr([t[0] for t in sn_t]), parr([t[4] for t in sn_t]), parr([t[5] for t in sn_t]),
                                            parr(sigl), parr([t[1] for t in sn_t]), p(work), st), 16 * tot, keep=(work, sig, sigl))
    add("spectral_norm_backward_multi(4 layers of D)", [len(SN), tot], lambda:
        lib.skd_spectral_norm_backward_multi(len(SN), hs, ws_, parr([t[0] for t in sn_t]), parr([t[4] for t in sn_t]), parr([t[5] for t in sn_t]),
                                             parr(sigl), parr([t[2] for t in sn_t]), parr([t[3] for t in sn_t]), p(work), st), 16 * tot)
    # pixel-wise KL (csrc/pixelwise.hip): read S, read T, write dS
    N, C, HW = 8, 19, 65 * 65
    S_, T_, dS = (torch.randn(N, C, HW, device=dev) for _ in range(3))
    pl, pws = torch.empty(1, device=dev), torch.empty(max(1, lib.skd_pixelwise_workspace_floats(N, HW)), device=dev)
    add("pixelwise_loss", [N, C, HW], lambda: lib.skd_pixelwise_loss(N, C, HW, p(S_), p(T_), p(pl), p(dS), p(pws), st), 12 * N * C * HW,
        keep=(S_, T_, dS, pl, pws))
    # fused bilinear upsample + CE of both heads (csrc/ce_dsn.hip): the int64 target + 2 x (logits read + gradient write)
    h_, w_, H_, W_ = 65, 65, 512, 512
    lm, ld, gm, gd = (torch.randn(N, C, h_, w_, device=dev) for _ in range(4))
    tg = torch.randint(0, C, (N, H_, W_), device=dev)
    cl_, cws = torch.empty(1, device=dev), torch.empty(max(8, lib.skd_ce_dsn_workspace_floats(N, C, h_, w_, H_, W_)), device=dev)
    add("ce_dsn_forward(+grads)", [N, C, h_, w_, H_, W_], lambda:
        lib.skd_ce_dsn_forward(N, C, h_, w_, H_, W_, p(lm), p(ld), p(tg), 255, 0.4, p(cl_), p(gm), p(gd), p(cws), st),
        8 * N * H_ * W_ + 4 * 4 * N * C * h_ * w_, keep=(lm, ld, gm, gd, tg, cl_, cws))
    # pyramid pooling, channels-last (csrc/ppm.hip): one read of the map (pool); one write of the concatenated tensor + reads (concat)
    sizes = [1, 2, 3, 6]
    sarr = iarr(sizes)
    for B_, Cf, Cm in ((8, 2048, 512), (8, 512, 128)):
        feats = torch.randn(B_, 65, 65, Cf, device=dev)
        pooled = torch.empty(lib.skd_ppm_pooled_floats(B_ * Cf, 4, sarr), device=dev)
        pws_ = torch.empty(max(1, lib.skd_ppm_nhwc_workspace_floats(B_, Cf, 0, 65, 65, 4, sarr)), device=dev)
        add("ppm_pool_nhwc", [B_, Cf, 65, 65], lambda B_=B_, Cf=Cf, feats=feats, pooled=pooled, pws_=pws_:
            lib.skd_ppm_pool_nhwc(B_, Cf, 65, 65, 4, sarr, p(feats), p(pooled), p(pws_), st), 4 * B_ * Cf * 65 * 65, keep=(feats, pooled, pws_))
        priors = [torch.randn(B_, s_, s_, Cm, device=dev) for s_ in sizes]
        cat = torch.empty(B_, 65, 65, 4 * Cm + Cf, device=dev)
        add("ppm_concat_nhwc", [B_, Cm, Cf, 65, 65], lambda B_=B_, Cf=Cf, Cm=Cm, priors=priors, feats=feats, cat=cat:
            lib.skd_ppm_concat_nhwc(B_, Cm, Cf, 65, 65, 4, sarr, parr(priors), p(feats), p(cat), st),
            4 * B_ * 65 * 65 * (Cf + 4 * Cm + Cf), keep=(priors, cat))
    # the stem's max-pool (csrc/maxpool.hip): 4 B/elem in + 5 B per output; backward 5 B per output + 4 B/elem out
    B_, C_, Hh, Ww, OH, OW = 8, 128, 256, 256, 129, 129
    mx, my, mdy, mdx = torch.randn(B_, Hh, Ww, C_, device=dev), torch.empty(B_, OH, OW, C_, device=dev), torch.randn(B_, OH, OW, C_, device=dev), torch.empty(B_, Hh, Ww, C_, device=dev)
    marg = torch.empty(B_, OH, OW, C_, dtype=torch.uint8, device=dev)
    add("maxpool3x3s2_nhwc", [B_, C_, Hh, Ww], lambda: lib.skd_maxpool3x3s2_nhwc(B_, C_, Hh, Ww, OH, OW, p(mx), p(my), p(marg), st),
        4 * B_ * C_ * Hh * Ww + 5 * B_ * C_ * OH * OW, keep=(mx, my, mdy, mdx, marg))
    add("maxpool3x3s2_backward_nhwc", [B_, C_, Hh, Ww], lambda: lib.skd_maxpool3x3s2_backward_nhwc(B_, C_, Hh, Ww, OH, OW, p(mdy), p(marg), p(mdx), st),
        4 * B_ * C_ * Hh * Ww + 5 * B_ * C_ * OH * OW)
    # round 6, the training stem fused (csrc/abn.hip "student stem"): forward = read x once, write pooled + argmax; backward reduce =
    # read x + (pooled gradient, argmax); backward dx = the same + write dx
    smean, svar = torch.zeros(C_, device=dev), torch.ones(C_, device=dev)
    sgam, sbet = torch.ones(C_, device=dev), torch.zeros(C_, device=dev)
    sws = torch.empty(max(1, lib.skd_abn_nhwc_workspace_floats(B_ * Hh * Ww, C_)), device=dev)
    sstat, sdw, sdb = torch.empty(2, C_, device=dev), torch.empty(C_, device=dev), torch.empty(C_, device=dev)
    add("abn_relu_maxpool3x3s2 (stem fwd, fused)", [B_, C_, Hh, Ww], lambda:
        lib.skd_abn_relu_maxpool3x3s2_nhwc(B_, C_, Hh, Ww, OH, OW, p(mx), p(smean), p(svar), p(sgam), p(sbet), 1e-5, p(my), p(marg), st),
        4 * B_ * C_ * Hh * Ww + 5 * B_ * C_ * OH * OW, keep=(smean, svar, sgam, sbet, sws, sstat, sdw, sdb))
    add("abn_relu_maxpool3x3s2_backward_reduce (stem bwd, fused)", [B_, C_, Hh, Ww], lambda:
        lib.skd_abn_relu_maxpool3x3s2_backward_reduce_nhwc(B_, C_, Hh, Ww, OH, OW, p(mx), p(mdy), p(marg), p(smean), p(svar), p(sgam), p(sbet),
                                                           p(sstat[0]), p(sstat[1]), 1e-5, p(sws), st),
        4 * B_ * C_ * Hh * Ww + 5 * B_ * C_ * OH * OW)
    add("abn_relu_maxpool3x3s2_backward_dx (stem bwd, fused)", [B_, C_, Hh, Ww], lambda:
        lib.skd_abn_relu_maxpool3x3s2_backward_dx_nhwc(B_, C_, Hh, Ww, OH, OW, p(mx), p(mdy), p(marg), p(smean), p(svar), p(sgam), p(sbet),
                                                       p(sstat[0]), p(sstat[1]), p(mdx), p(sdw), p(sdb), 1e-5, 0, st),
        8 * B_ * C_ * Hh * Ww + 5 * B_ * C_ * OH * OW)
    # ... and the un-fused sequence it replaces on the same tensors: normalise + ReLU, (pool above), (un-pool above), reduce, dx
    sy = torch.empty(B_, Hh, Ww, C_, device=dev)
    add("abn_apply_nhwc_to(relu) (stem fwd, un-fused part)", [B_ * Hh * Ww, C_], lambda:
        lib.skd_abn_apply_nhwc_to(B_ * Hh * Ww, C_, p(mx), None, p(sy), p(smean), p(svar), p(sgam), p(sbet), 1e-5, 3, 0.0, st),
        8 * B_ * C_ * Hh * Ww, keep=(sy,))
    add("abn_relu_backward_reduce_nhwc_x (stem bwd, un-fused part)", [B_ * Hh * Ww, C_], lambda:
        lib.skd_abn_relu_backward_reduce_nhwc_x(B_ * Hh * Ww, C_, p(mx), p(sy), p(smean), p(svar), p(sgam), p(sbet), p(sstat[0]), p(sstat[1]), 1e-5,
                                                p(sws), st), 8 * B_ * C_ * Hh * Ww)
    add("abn_relu_backward_dx_nhwc_x (stem bwd, un-fused part)", [B_ * Hh * Ww, C_], lambda:
        lib.skd_abn_relu_backward_dx_nhwc_x(B_ * Hh * Ww, C_, p(mx), p(sy), p(smean), p(svar), p(sgam), p(sbet), p(sstat[0]), p(sstat[1]), p(mdx),
                                            p(sdw), p(sdb), 1e-5, 0, st), 12 * B_ * C_ * Hh * Ww)
    # the pair-wise criterion's max-pool with argmax (csrc/pairwise.hip, NCHW planes): read the feature maps once
    for planes, kh in ((8 * 128, 32), (8 * 512, 32), (8 * 512, 1)):
        fx = torch.randn(planes, 65, 65, device=dev)
        oh = -(-65 // kh)
        po, pi = torch.empty(planes, oh, oh, device=dev), torch.empty(planes, oh, oh, dtype=torch.int32, device=dev)
        add("maxpool_argmax(pair-wise pooling)", [planes, 65, 65, kh], lambda planes=planes, kh=kh, fx=fx, po=po, pi=pi:
            lib.skd_maxpool_argmax(planes, 65, 65, kh, kh, p(fx), p(po), p(pi), st), 4 * planes * 65 * 65 + 8 * planes * oh * oh, keep=(fx, po, pi))
    # ... and its channels-last form (round 5: what the step runs -- the PSP features are pooled as they are) + the un-pool
    for Bp, Cp, kh in ((8, 128, 32), (8, 512, 32), (8, 512, 1)):
        fx = torch.randn(Bp, 65, 65, Cp, device=dev)
        oh = -(-65 // kh)
        po, pi = torch.empty(Bp * Cp, oh * oh, device=dev), torch.empty(Bp * Cp, oh * oh, dtype=torch.int32, device=dev)
        add("maxpool_argmax_nhwc(pair-wise pooling)", [Bp, Cp, 65, 65, kh], lambda Bp=Bp, Cp=Cp, kh=kh, fx=fx, po=po, pi=pi:
            lib.skd_maxpool_argmax_nhwc(Bp, Cp, 65, 65, kh, kh, p(fx), p(po), p(pi), st), 4 * Bp * Cp * 65 * 65 + 8 * Bp * Cp * oh * oh, keep=(fx, po, pi))
        if Cp == 128:
            dpo, dxx = torch.randn(Bp * Cp, oh * oh, device=dev), torch.empty(Bp, 65, 65, Cp, device=dev)
            add("maxunpool_scatter_nhwc", [Bp, Cp, 65, 65, kh], lambda Bp=Bp, Cp=Cp, kh=kh, dpo=dpo, pi=pi, dxx=dxx, oh=oh:
                lib.skd_maxunpool_scatter_nhwc(Bp, Cp, 65, 65, kh, kh, p(dpo), oh * oh, p(pi), p(dxx), st), 4 * Bp * Cp * 65 * 65 + 8 * Bp * Cp * oh * oh,
                keep=(dpo, dxx))
    # round 6: the 19-class 1x1 heads (csrc/head.hip): forward reads the map once, writes the logits; backward reads map + logit
    # gradient, writes the map's gradient
    for Bh, Kh in ((8, 128), (8, 512)):
        hx, hwt, hb = torch.randn(Bh * 65 * 65, Kh, device=dev), torch.randn(19, Kh, device=dev) * 0.05, torch.randn(19, device=dev)
        ho, hg = torch.empty(Bh, 19, 65 * 65, device=dev), torch.randn(Bh, 19, 65 * 65, device=dev)
        add("head1x1_forward", [Bh, 65 * 65, Kh, 19], lambda Bh=Bh, Kh=Kh, hx=hx, hwt=hwt, hb=hb, ho=ho:
            lib.skd_head1x1_forward_nhwc(Bh, 65 * 65, Kh, 19, p(hx), p(hwt), p(hb), p(ho), st), 4 * Bh * 65 * 65 * (Kh + 19), keep=(hx, hwt, hb, ho, hg))
        if Kh == 128:
            hgx, hgw, hgb = torch.empty_like(hx), torch.empty_like(hwt), torch.empty_like(hb)
            hws = torch.empty(max(1, lib.skd_head1x1_backward_workspace_floats(Bh, 65 * 65, Kh, 19)), device=dev)
            add("head1x1_backward", [Bh, 65 * 65, Kh, 19], lambda Bh=Bh, Kh=Kh, hx=hx, hwt=hwt, hg=hg, hgx=hgx, hgw=hgw, hgb=hgb, hws=hws:
                lib.skd_head1x1_backward_nhwc(Bh, 65 * 65, Kh, 19, p(hx), p(hwt), p(hg), p(hgx), p(hgw), p(hgb), p(hws), st),
                4 * Bh * 65 * 65 * (2 * Kh + 19), keep=(hgx, hgw, hgb, hws))
    # evaluation tail (csrc/evaluate.hip): 8 B label + 1 B prediction per pixel, the 129 x 257 logits from cache
    el = torch.randn(1, 19, 129, 257, device=dev)
    elab = torch.randint(0, 19, (1, 1024, 2048), device=dev)
    econf, epred = torch.zeros(19, 19, dtype=torch.int64, device=dev), torch.empty(1, 1024, 2048, dtype=torch.uint8, device=dev)
    add("seg_confusion(1024x2048)", [1, 19, 129, 257, 1024, 2048], lambda:
        lib.skd_seg_confusion(1, 19, 129, 257, 1024, 2048, p(el), p(elab), 255, p(epred), p(econf), st), 9 * 1024 * 2048, keep=(el, elab, econf, epred))
    # the frozen bottleneck's fused tail GEMM (csrc/conv1x1.hip) at the teacher's four problem shapes (super-tile order; the panel-major
    # A/B of round 4 is recorded in profiles/r04g_pmc.json).  Algorithmic bytes: X (M x K) + W (N x K) + residual (M x N) read, Y written.
    for M_, K_, N_ in ((8 * 129 * 129, 64, 256), (8 * 65 * 65, 128, 512), (8 * 65 * 65, 256, 1024), (8 * 65 * 65, 512, 2048)):
        X_, Wt_ = torch.randn(M_, K_, device=dev), torch.randn(N_, K_, device=dev) * 0.05
        R_, Y_ = torch.randn(M_, N_, device=dev), torch.empty(M_, N_, device=dev)
        mu, va, ga, be = torch.zeros(N_, device=dev), torch.ones(N_, device=dev), torch.ones(N_, device=dev), torch.zeros(N_, device=dev)
        pm, pv = torch.zeros(K_, device=dev), torch.ones(K_, device=dev)
        pack = torch.empty(4 * K_, device=dev)
        assert lib.skd_abn_pack_eval_params(K_, p(pm), p(pv), None, None, 1e-5, p(pack), st)
        keep = (X_, Wt_, R_, Y_, mu, va, ga, be, pm, pv, pack)

        def gemm(M_=M_, K_=K_, N_=N_, X_=X_, Wt_=Wt_, R_=R_, Y_=Y_, mu=mu, va=va, ga=ga, be=be, pack=pack):
            return lib.skd_conv1x1_abn_pro_nhwc(M_, K_, N_, p(X_), p(Wt_), p(R_), p(Y_), p(mu), p(va), p(ga), p(be), 1e-5, p(pack), 3, 0.0, st)
        nbytes = 4 * (M_ * K_ + N_ * K_ + 2 * M_ * N_)
        add("conv1x1_abn_pro(tail GEMM, super-tile order)", [M_, K_, N_], gemm, nbytes, flops=2.0 * M_ * K_ * N_, keep=keep)
    return cases


def main():
    import torch
    from structure_knowledge_distillation_amd import _lib
    lib = _lib.load()
    mode = sys.argv[1] if len(sys.argv) > 1 else "time"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    only = sys.argv[3] if len(sys.argv) > 3 else None
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    cases = build_cases(lib, torch, dev, st)
    print(json.dumps({"manifest": [{k: c[k] for k in ("id", "name", "shape", "algo_bytes", "algo_flops")} for c in cases]}), flush=True)
    marker_buf = torch.zeros(256 * (len(cases) + 2), device=dev)
    torch.cuda.synchronize()
    for c in cases:
        if only and only not in c["name"]:
            continue
        fn = c["fn"]
        if mode == "pmc":
            assert lib.skd_leaky_relu(256 * (c["id"] + 1), marker_buf.data_ptr(), 1.0, st)     # marker: grid = id + 1 workgroups
            for _ in range(2):
                assert fn()
            continue
        assert fn()
        ms = warm_timed(fn, reps, warm_ms=15.0)          # tools/_timing.py: warm clocks, median of five groups
        row = {"id": c["id"], "name": c["name"], "shape": c["shape"], "us": round(ms * 1e3, 2)}
        if c["algo_bytes"]:
            row["GBs"] = round(c["algo_bytes"] / (ms * 1e-3) / 1e9, 1)
            row["frac_hbm_8TBs"] = round(row["GBs"] / 8000.0, 3)
        if c["algo_flops"]:
            tf = round(c["algo_flops"] / (ms * 1e-3) / 1e12, 2)
            if c["name"].startswith("pairwise"):      # symmetric Gram: only the upper-triangle tiles execute -- a convention, not a utilisation
                row["TFLOPs(full-matrix convention)"] = tf
            else:
                row["TFLOPs"] = tf
                row["frac_fp32_mfma_157.3"] = round(tf / 157.3, 4)
        print(json.dumps(row), flush=True)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
