"""GPU: the C-ABI launches are stream-ordered, allocation-free inside the library and never synchronise, so the
HIP side of a leapfrog step can be captured into a hipGraph and replayed with identical results."""
import pytest
import torch

from oracle import operators as oops, schedule as osched

pytestmark = pytest.mark.gpu


def test_hot_path_step_is_graph_capturable():
    import nhmc.kernels as K
    from nhmc import operators
    B, dim = 4, 64
    g_ = torch.Generator().manual_seed(5)
    op = operators.Inpainting(3, dim, oops.random_inpaint_missing(dim, generator=g_), 'cuda')
    x0 = torch.randn(B, 3, dim, dim, generator=g_).cuda()
    p0 = torch.randn(B, 3, dim, dim, generator=g_).cuda()
    e = torch.randn(B, 6, dim, dim, generator=g_).cuda()
    y = torch.randn(B, op.M, generator=g_).cuda()
    b = osched.betas_fp32()
    at = [osched.alpha_bar(b, torch.full((B,), t)).reshape(-1).cuda() for t in (750, 500, 250)]
    atn = [osched.alpha_bar(b, torch.full((B,), t)).reshape(-1).cuda() for t in (500, 250, -1)]
    eps = torch.full((B,), 0.05, dtype=torch.float64, device='cuda')
    sig = torch.full((B,), 0.9, dtype=torch.float64, device='cuda')

    def step(x, p):
        cur, ins = x, []
        for s in range(3):
            ins.append(cur)
            cur = K.ddim_mix_fwd(cur, e, at[s], atn[s], final_clip=(s == 2))['xt_next']
        loss, g, _ = op.fused_last_vjp(ins[2], e, at[2], atn[2], y)
        for s in (1, 0):
            g, _ = K.ddim_mix_bwd(g, ins[s], e, at[s], atn[s])
        K.leapfrog_fused(K.LF_MID, x, p, g, eps, sig, 1.0)
        return loss

    xe, pe = x0.clone(), p0.clone()
    loss_e = step(xe, pe)                                   # eager
    xs, ps = x0.clone(), p0.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                           # warm-up on the capture stream (allocator pools)
        step(xs.clone(), ps.clone())
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        loss_g = step(xs, ps)
    xs.copy_(x0)
    ps.copy_(p0)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(xs, xe) and torch.equal(ps, pe) and torch.equal(loss_g, loss_e)
    graph.replay()                                          # a second replay advances the state again
    torch.cuda.synchronize()
    xe2, pe2 = xe.clone(), pe.clone()
    step(xe2, pe2)
    assert torch.equal(xs, xe2) and torch.equal(ps, pe2)


class SmallScore(torch.nn.Module):
    """A capturable score stand-in (no host->device copies in forward, unlike the oracle's TinyScore)."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a = torch.nn.Conv2d(3, 8, 3, padding=1)
        self.b = torch.nn.Conv2d(8, 6, 3, padding=1)

    def forward(self, x, t):
        return self.b(torch.tanh(self.a(x)) * (1.0 + t.view(-1, 1, 1, 1) / 1000.0))


def test_engine_graph_replay_matches_eager():
    """decode+gradient of a chunk (score network included) replayed as a hipGraph == eager launches."""
    from nhmc import operators, plugin, sampler
    dim, B = 32, 3
    g_ = torch.Generator().manual_seed(11)
    op = operators.Inpainting(3, dim, oops.random_inpaint_missing(dim, generator=g_), 'cuda')
    algo = plugin.HMC(SmallScore().cuda().requires_grad_(False), op, 0.1)
    eng = sampler.LeapfrogEngine(algo.score, op, osched.betas_fp32().cuda(), [250, 500, 750], [-1, 250, 500],
                                 torch.device('cuda'), chunk=2)                 # ragged: chunks of 2 and 1 -> two graphs
    x = torch.randn(B, 3, dim, dim, generator=g_).cuda()
    y = torch.randn(B, op.M, generator=g_).cuda()
    eager = eng.decode_and_grad(x, y)
    for _ in range(2):                                                           # capture, then a pure replay
        graphed = eng.decode_and_grad(x, y, graph=True)
        for a, b in zip(eager, graphed):
            assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-12
    x2 = x * 0.5                                                                 # new inputs through the same graphs
    a = eng.decode_and_grad(x2, y)
    b = eng.decode_and_grad(x2, y, graph=True)
    assert all(float((u - v).abs().max()) <= 1e-5 * float(u.abs().max()) + 1e-12 for u, v in zip(a, b))
    assert len(eng._graphs) == 2
