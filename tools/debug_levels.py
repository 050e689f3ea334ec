"""Development tool (GPU box): where do the hand-off levels differ from the recompute backward?"""
import sys, torch
sys.path.insert(0, ".")
import mentflow_amd as mf
dev = torch.device("cuda", 0)

def run(d, bins, L, n, level, T=1):
    torch.manual_seed(0)
    gen = mf.generate.build_generator("nsf", device=dev, input_features=d, output_features=d, hidden_layers=L, hidden_units=64, transforms=T, bins=bins)
    with torch.no_grad():
        for layer in gen.layers:
            lin = layer.linears()[-1]; lin.weight.mul_(4.0); lin.bias.add_(torch.randn_like(lin.bias))
    gen.spec().act_level = level
    torch.manual_seed(1)
    z = (torch.randn(n, d) * 1.5).to(dev).requires_grad_(True)
    wx, wl = torch.randn(n, d).to(dev), torch.randn(n).to(dev)
    x, lp = gen.sample_and_log_prob(n, z=z)
    ((x * wx).sum() + (lp * wl).sum()).backward()
    return gen, x.detach(), lp.detach(), z.grad

for (d, bins, L, n) in [(6, 20, 3, 128), (6, 20, 3, 700), (6, 20, 3, 42705), (6, 8, 3, 700), (6, 20, 2, 700), (3, 20, 3, 700)]:
    g0, x0, l0, z0 = run(d, bins, L, n, 0)
    for level in (1, 2):
        g, x, l, zg = run(d, bins, L, n, level)
        print(f"d={d} bins={bins} L={L} n={n} level {level}: x {float((x-x0).abs().max()):.2e} logp {float((l-l0).abs().max()):.2e} dz {float((zg-z0).abs().max()):.2e}"
              f" (rows differing: {int(((zg-z0).abs().sum(1) > 0).sum())})")
        for (nm, p), (_, q) in zip(g.named_parameters(), g0.named_parameters()):
            dd = (p.grad - q.grad).abs()
            if float(dd.max()) > 0:
                print(f"     {nm:60s} {tuple(p.shape)} max diff {float(dd.max()):.3e} of {float(q.grad.abs().max()):.3e}; {int((dd > 0).sum())} entries")
