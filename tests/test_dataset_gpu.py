"""Data plane (SURVEY f-4): the streamed (pinned-host, double-buffered async H2D) mode yields exactly the batches of the
HBM-resident mode, and a training loop that consumes it on its own stream sees each batch intact."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_streamed_batches_equal_resident_batches():
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd.gan.dataset import GANDataset
    a = GANDataset.synthetic(203, 32, 8, 16, seed=5, resident=True)
    b = GANDataset.synthetic(203, 32, 8, 16, seed=5, resident=False)
    assert b.notes.is_pinned() and not b.notes.is_cuda and a.notes.is_cuda
    stream = torch.cuda.Stream()
    sink = torch.empty(16, 32, 8, device="cuda")
    for epoch in range(2):
        ga, gb = torch.Generator().manual_seed(epoch), torch.Generator().manual_seed(epoch)
        n = 0
        with torch.cuda.stream(stream):
            for (xa, na, la, ea), (xb, nb, lb, eb) in zip(a.batches(16, ga), b.batches(16, gb)):
                sink.copy_(xb)                       # the consumer's use of the staged batch, on ITS stream
                busy = sink @ sink.transpose(1, 2)   # noqa: F841  keep the stream busy while the next copy is issued
                assert torch.equal(xa, xb) and torch.equal(na, nb) and torch.equal(la, lb) and torch.equal(ea, eb)
                n += 1
        assert n == 203 // 16


def test_stage_rows_gathers_and_copies_in_one_launch():
    """mg_stage_rows against torch.index_select: 16-byte rows, 4-byte rows (int64 labels, odd widths), an unindexed
    job, a misaligned source view, and out-of-range indices (clamped, never read outside the source)."""
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd import ops
    g = torch.Generator().manual_seed(0)
    notes = torch.rand(37, 8, 12, generator=g).cuda()
    odd = torch.rand(38, 7, generator=g).cuda()[1:]                 # rows of 28 bytes starting 28 bytes into the allocation
    lab = torch.randint(0, 4, (37,), generator=g).cuda()
    dense = torch.rand(16, 5, generator=g).cuda()
    idx = torch.randint(0, 37, (16,), generator=g).cuda()
    out = [torch.full((16, 8, 12), -1.0).cuda(), torch.full((16, 7), -1.0).cuda(), torch.full((16,), -1).cuda(),
           torch.full((20, 5), -1.0).cuda()]
    ops.stage_rows([(notes, out[0], idx), (odd, out[1], idx), (lab, out[2], idx), (dense, out[3], None)], 16)
    assert torch.equal(out[0], notes.index_select(0, idx)) and torch.equal(out[1], odd.index_select(0, idx))
    assert torch.equal(out[2], lab.index_select(0, idx))
    assert torch.equal(out[3][:16], dense) and bool((out[3][16:] == -1).all())
    bad = idx.clone()
    bad[0], bad[1] = -5, 10 ** 12
    ops.stage_rows([(notes, out[0], bad)], 16)
    assert torch.equal(out[0][0], notes[0]) and torch.equal(out[0][1], notes[36]) and torch.equal(out[0][2:], notes[idx[2:]])
    # column blocks of a wider matrix (a row pitch between destination rows): [odd | dense | untouched]
    wide = torch.full((16, 7 + 5 + 3), -2.0).cuda()
    ops.stage_rows([(odd, wide[:, :7], idx), (dense, wide[:, 7:12], None)], 16)
    assert torch.equal(wide[:, :7], odd.index_select(0, idx)) and torch.equal(wide[:, 7:12], dense)
    assert bool((wide[:, 12:] == -2).all())
    with pytest.raises(ValueError):
        ops.stage_rows([(notes, out[1], idx)], 16)                  # row shapes differ
    with pytest.raises(ValueError):
        ops.stage_rows([(dense, out[3], None)], 20)                 # unindexed source shorter than the batch


def test_batch_stage_fills_the_engine_like_set_batch():
    """Batch.stage (one gather launch from the resident arrays) leaves the engine's inputs exactly as set_batch on the
    four gathered tensors does, in both dataset modes."""
    import melo_gan_amd  # noqa: F401
    from oracle import melo_oracle as O
    from melo_gan_amd.gan.dataset import GANDataset
    from melo_gan_amd.gan.engine import GanEngine
    B, T, Cn = 8, 32, 4
    cfg = O.default_gan_cfg(B, T, Cn)
    LAT = cfg["LATENT_DIM"]
    eng = GanEngine(cfg, O.default_ed_cfg(Cn), "cuda", B)
    for resident in (True, False):
        ds = GANDataset.synthetic(50, T, Cn, LAT, seed=2, resident=resident)
        seen = 0
        for batch in ds.batches(B, torch.Generator().manual_seed(1)):
            notes, numeric, latent, emot = batch.tensors()
            for t in (eng.X0, eng.numeric, eng.latent):
                t.fill_(float("nan"))
            eng.emot_idx.fill_(-1)
            batch.stage(eng)
            got = [eng.real.clone(), eng.numeric.clone(), eng.latent.clone(), eng.emot_idx.clone()]
            eng.set_batch(notes.cpu(), numeric.cpu(), latent.cpu(), emot.cpu())          # host sources: torch copies
            want = [eng.real, eng.numeric, eng.latent, eng.emot_idx]
            assert all(torch.equal(a, b) for a, b in zip(got, want))
            assert torch.equal(got[0], notes) and torch.equal(got[3], emot)
            seen += 1
        assert seen == 50 // B


def test_bound_split_stages_its_batches_inside_the_step():
    """GanEngine.bind_batches / GANDataset.bind: the first launch of every batch's graph gathers batch k of the epoch's
    order on the device (the Philox step counter is the batch counter) -- equal to set_batch(idx) of the same indices, under
    graph replay, across an epoch change, for critic-only and critic+generator batches."""
    import melo_gan_amd  # noqa: F401
    from melo_gan_amd.gan.dataset import GANDataset
    from melo_gan_amd.gan.engine import GanEngine
    from melo_gan_amd.gan.dp import DataParallel
    from oracle import melo_oracle as O
    B, T, C, N = 4, 32, 4, 22
    cfg, ed_cfg = O.default_gan_cfg(B, T, C), O.default_ed_cfg(C)
    ds = GANDataset.synthetic(N, T, C, cfg["LATENT_DIM"], 3, "cuda")
    engs = []
    for _ in range(2):
        e = GanEngine(cfg, ed_cfg, "cuda", B)
        e.init_weights(1)
        e.seed(11)
        engs.append(e)
    bound, plain = engs
    dpb, dpp = DataParallel(bound, 1, None), DataParallel(plain, 1, None)
    gen_b, gen_p = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
    with torch.cuda.stream(bound.stream):
        nb = ds.bind(bound, B)
        assert nb == N // B
        for epoch in range(3):
            assert ds.start_epoch(bound, gen_b) == nb
            batches = list(ds.batches(B, gen_p))
            for k in range(nb):
                g_step = k % 2 == 1
                dpb.step(True, g_step)
                batches[k].stage(plain)
                dpp.step(True, g_step)
                torch.cuda.synchronize()
                assert torch.equal(bound.real, plain.real) and torch.equal(bound.numeric, plain.numeric) and torch.equal(bound.numeric_d, plain.numeric_d)
                assert torch.equal(bound.emot_idx, plain.emot_idx) and torch.equal(bound.latent, plain.latent)
        torch.cuda.synchronize()
    assert torch.equal(bound.D.data, plain.D.data) and torch.equal(bound.GE.data, plain.GE.data)
    assert int(bound.rng_step.item()) == 3 * nb
