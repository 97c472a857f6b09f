"""Debug aid: per-parameter gradient errors of the APF encoder against the rounding-matched statement."""
import sys, torch
sys.path.insert(0, '.')
import p4p_amd
from p4p_amd import apf as PA
from oracle import apf_oracle as A
from tests.emulate import apf_encoder_emulated

def rel_l2(a, b):
    a, b = a.detach().double().flatten().cpu(), b.detach().double().flatten().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))

for G, ec in [(48, 128), (40, 768), (400, 128)]:
    gs = 32
    m = PA.Encoder(ec, 6)
    m.load_state_dict(A.seeded_state_dict(A.Encoder(ec, 6), seed=ec))
    m = m.to('cuda').train()
    x = (torch.randn(1, G, gs, 6, generator=torch.Generator().manual_seed(5)) * 0.3).to('cuda')
    from p4p_amd import functional_apf as FA
    FA._CAPTURE = {}
    tok = m(x)
    gates, FA._CAPTURE = FA._CAPTURE, None
    w = torch.randn(tok.shape, generator=torch.Generator().manual_seed(6)).to('cuda')
    (tok * w).sum().backward()
    params = {n: p.detach().clone().requires_grad_(True) for n, p in m.named_parameters()}
    te = apf_encoder_emulated(x.reshape(G * gs, 6), params, G, gs, gates=gates)
    (te * w.view(G, ec)).sum().backward()
    print(G, ec, 'fwd', rel_l2(tok.reshape(G, ec), te), 'frac exact', float((tok.reshape(G, ec) == te).float().mean()))
    for n, p in m.named_parameters():
        if params[n].grad is None:
            continue
        print('   %-28s %.4g   |g|=%.3g |ref|=%.3g' % (n, rel_l2(p.grad, params[n].grad), float(p.grad.norm()), float(params[n].grad.norm())))
