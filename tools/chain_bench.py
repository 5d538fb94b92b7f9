"""The row chains of the cfg2 step in isolation (hipGraph replay of 20 back-to-back launches each), against the per-layer
launches they replace (MELO_CHAINS=0 engine): numeric encoder forward (2B rows) / data-gradient, critic tail (3B and B rows),
emotion classifier tail."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, melo_gan_amd  # noqa
from melo_gan_amd import ops
from melo_gan_amd.gan.engine import GanEngine
from melo_gan_amd.gan.config import default_gan_cfg, default_ed_cfg
from _timeit import timeit

B, T, C = 64, 256, 128


def make(chains):
    os.environ["MELO_CHAINS"] = "1" if chains else "0"
    e = GanEngine(default_gan_cfg(B, T, C), default_ed_cfg(C), "cuda", B)
    os.environ.pop("MELO_CHAINS")
    e.init_weights(1)
    g = torch.Generator().manual_seed(0)
    e.set_batch((torch.rand(B, T, C, generator=g) * 2 - 1).cuda(), torch.randn(B, 6, generator=g).cuda(), torch.zeros(B, e.latent_dim).cuda(),
                torch.randint(0, 4, (B,), generator=g).cuda())
    with torch.cuda.stream(e.stream):
        e.draw_randoms_both()
        e.dg_forward(); e.d_backward(forward=False); e.d_update(); e.g_backward_a2(); e.g_backward_b(); e.g_update()
    torch.cuda.synchronize()
    return e


def ed_tail(e):
    if e._chain_ed:
        # the chain part only: skip the convolutions by timing the whole branch minus them is not possible; rebuild the chain here
        P = e.ED.p
        ch = ops.Chain(B)
        ch.mean_t(0, e.ed_a[3], out=e.ed_pool)
        ch.linear_fwd(0, 1, P["encoder.project.weight"], P["encoder.project.bias"], out=e.ed_proj)
        ch.linear_fwd(1, 2, P["classifier.net.0.weight"], P["classifier.net.0.bias"], ops.ACT_GELU, zout=e.ed_cz[0], out=e.ed_ca[0])
        ch.linear_fwd(2, 3, P["classifier.net.3.weight"], P["classifier.net.3.bias"], ops.ACT_GELU, zout=e.ed_cz[1], out=e.ed_ca[1])
        ch.linear_fwd(3, 4, P["classifier.head.weight"], P["classifier.head.bias"], out=e.logits)
        ch.softmax_ce(4, 5, e.emot_idx, e.ed_loss_rows, e.lambda_emo / B, 4)
        ch.store(5, e.dlogits)
        ch.linear_dgrad(5, 0, P["classifier.head.weight"], gref=e.ed_cz[1], gact=ops.ACT_GELU, out=e.ed_dcz[1])
        ch.linear_dgrad(0, 1, P["classifier.net.3.weight"], gref=e.ed_cz[0], gact=ops.ACT_GELU, out=e.ed_dcz[0])
        ch.linear_dgrad(1, 2, P["classifier.net.0.weight"], out=e.ed_dproj)
        ch.linear_dgrad(2, 3, P["encoder.project.weight"], out=e.ed_dpool)
        ch.launch()
    else:
        P = e.ED.p
        ops.meanT_fwd(e.ed_a[3], e.ed_pool)
        ops.linear_fwd(e.ed_pool, P["encoder.project.weight"], e.ed_proj, bias=P["encoder.project.bias"])
        feat = e.ed_proj
        for j in range(2):
            ops.linear_fwd(feat, P[f"classifier.net.{3 * j}.weight"], e.ed_ca[j], bias=P[f"classifier.net.{3 * j}.bias"], zout=e.ed_cz[j], act=ops.ACT_GELU)
            feat = e.ed_ca[j]
        ops.linear_fwd(feat, P["classifier.head.weight"], e.logits, bias=P["classifier.head.bias"])
        ops.softmax_ce(e.logits, e.emot_idx, e.emo, e.dlogits, e.lambda_emo)
        g, w = e.dlogits, P["classifier.head.weight"]
        for j in (1, 0):
            ops.linear_dgrad(g, w, e.ed_dcz[j], gref=e.ed_cz[j], gact=ops.ACT_GELU)
            g, w = e.ed_dcz[j], P[f"classifier.net.{3 * j}.weight"]
        ops.linear_dgrad(g, w, e.ed_dproj)
        ops.linear_dgrad(e.ed_dproj, P["encoder.project.weight"], e.ed_dpool)


def e_bwd(e):
    PE = e._ep
    if e._chain_e:
        ch = ops.Chain(B)
        ch.load(0, e.d_gin[:, e.noise_dim:e.noise_dim + e.E]); ch.load(0, e.demb, accumulate=True); ch.store(0, e.demb)
        ch.linear_dgrad(0, 1, PE("net.7.weight"), gref=e.e_z2, gact=ops.ACT_GELU, mask=e.dmask[1], out=e.d_ez2)
        ch.linear_dgrad(1, 2, PE("net.4.weight"), gref=e.e_z1, gact=ops.ACT_GELU, mask=e.dmask[0], out=e.d_ez1)
        ch.linear_dgrad(2, 3, PE("net.1.weight"), out=e.d_ex0)
        ch.launch()
    else:
        ops.copy_cols(e.d_gin, e.noise_dim, e.demb, 0, e.E, accumulate=True)
        ops.linear_dgrad(e.demb, PE("net.7.weight"), e.d_ez2, gref=e.e_z2, gact=ops.ACT_GELU, emul=e.dmask[1])
        ops.linear_dgrad(e.d_ez2, PE("net.4.weight"), e.d_ez1, gref=e.e_z1, gact=ops.ACT_GELU, emul=e.dmask[0])
        ops.linear_dgrad(e.d_ez1, PE("net.1.weight"), e.d_ex0)


def d_tail(e, nb, ds, demb):
    P = e.D.p
    if not e._chain_d:
        ops.linear_fwd(e.H[:nb], P["fc.1.weight"], e.Fh[:nb], bias=P["fc.1.bias"], act=ops.ACT_LRELU)
    # _d_bwd_input up to dH (the chain, or dhead_fwd_bwd + linear_dgrad)
    if e._chain_d:
        ch = ops.Chain(nb)
        ch.load(0, e.H[:nb]); ch.linear_fwd(0, 1, P["fc.1.weight"], P["fc.1.bias"], ops.ACT_LRELU, out=e.Fh[:nb])
        ch.dhead(1, 2, P["real_fake.weight"].view(-1), P["real_fake.bias"], e.emb_d, ds, e.s[:nb], demb=demb)
        ch.store(2, e.dU[:nb]); ch.linear_dgrad(2, 3, P["fc.1.weight"], out=e.dH[:nb]); ch.launch()
    else:
        ops.dhead_fwd_bwd(ds, e.Fh[:nb], e.emb_d, P["real_fake.weight"].view(-1), P["real_fake.bias"], e.s[:nb], e.dU[:nb], demb,
                          nb_emb=nb if demb is not None else 0)
        ops.linear_dgrad(e.dU[:nb], P["fc.1.weight"], e.dH[:nb])


for chains in (False, True):
    e = make(chains)
    tag = "chain     " if chains else "per-layer "
    print(f"{tag} encoder fwd (2B rows + gin): {timeit(lambda: e._e_fwd(True, 'both', gin=True)):7.1f} us"
          + ("" if chains else f"  (+ stage_rows {timeit(lambda: ops.stage_rows([(e.noise_2, e.gin_2[:, :128], None), (e.emb_2, e.gin_2[:, 128:256], None)], 2 * B)):5.1f})"), flush=True)
    e._gin_done = False
    print(f"{tag} encoder data-gradient      : {timeit(lambda: e_bwd(e)):7.1f} us", flush=True)
    print(f"{tag} critic tail, 3B rows       : {timeit(lambda: d_tail(e, 3 * B, e.ds_d, None)):7.1f} us", flush=True)
    print(f"{tag} critic tail, B rows (+demb): {timeit(lambda: d_tail(e, B, e.ds_g, e.demb)):7.1f} us", flush=True)
    print(f"{tag} emotion classifier tail    : {timeit(lambda: ed_tail(e)):7.1f} us", flush=True)
