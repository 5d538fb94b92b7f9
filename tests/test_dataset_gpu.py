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
