"""Test helper: a plain PyTorch fp32 statement of P3Embed's local MLP that rounds activations to bf16 at
the same places as the HIP pipeline (F1, H2, F2, H3 and the GEMM operands).  ReLU masks and max-pool
arg-max positions are then identical to the kernels', which separates kernel-logic errors (must agree
to ~1%) from the hard-gating flips that any bf16 activation storage causes against an fp32 oracle."""
import torch


def _r(x):
    """round to bf16, straight-through gradient"""
    return x + (x.to(torch.bfloat16).float() - x).detach()


def _bn_train(h, gamma, beta, eps):
    mean = h.mean(0)
    var = h.var(0, unbiased=False)
    return mean, var, gamma / torch.sqrt(var + eps), beta - mean * gamma / torch.sqrt(var + eps)


def p3embed_emulated(points, idx, params, k, eps=1e-5):
    """points (B,N,3) f32 cuda, idx (B,S,k) int32; params = dict of fp32 leaf tensors named like the module.
    Returns tokens [B*S, 256] (fp32) computed the way csrc/embed.hip + the GEMM epilogues do."""
    B, N, _ = points.shape
    S = idx.shape[1]
    b = torch.arange(B, device=points.device).view(B, 1, 1)
    g = points[b, idx.long()]                                    # (B,S,k,3)
    x6 = torch.cat([g, g], dim=-1).reshape(B * S * k, 6)
    W1, W2, b2 = params['0.0.weight'].reshape(256, 6), params['0.1.weight'].reshape(256, 256), params['0.1.bias']
    aeff = W2 @ W1
    h1 = x6 @ aeff.t() + b2
    _, _, sc1, sh1 = _bn_train(h1, params['0.2.weight'], params['0.2.bias'], eps)
    f1 = _r(torch.relu(h1 * sc1 + sh1))                          # F1 stored bf16
    pooled = f1.view(B * S, k, 256).max(1).values
    W3 = params['1.0.weight'].reshape(512, 512)
    u = pooled @ _r(W3[:, :256]).t()
    h2 = f1 @ _r(W3[:, 256:]).t() + u.repeat_interleave(k, 0)
    _, _, sc2, sh2 = _bn_train(h2, params['1.1.weight'], params['1.1.bias'], eps)
    f2 = _r(torch.relu(_r(h2) * sc2 + sh2))
    h3 = f2 @ _r(params['1.3.weight'].reshape(256, 512)).t()
    _, _, sc3, sh3 = _bn_train(h3, params['1.4.weight'], params['1.4.bias'], eps)
    return torch.relu(_r(h3) * sc3 + sh3).view(B * S, k, 256).max(1).values


def apf_encoder_emulated(x6, params, G, gs, eps=1e-5, gates=None):
    """x6 [G*gs, 6] f32; params = dict of fp32 leaf tensors named like Encoder's state dict (first_conv.N.* /
    second_conv.N.*).  Tokens [G, E] computed the way functional_apf._ApfEncoderFn does: bf16 storage of H1, F1, H2, F2,
    f, H4, F4, H5 and of the GEMM weight operands; BatchNorm statistics from the fp32 accumulators; conv biases in front
    of a train-mode BatchNorm dropped (they cancel).  ``gates`` (functional_apf._CAPTURE of the kernel run) replaces every
    ReLU / arg-max decision by the kernels' own, so that the two backward passes differ by arithmetic only: with bf16
    storage in front of a 32-way max, one-ulp differences would otherwise move ~0.3 % of the arg-max picks."""
    p = params
    E = p['first_conv.6.weight'].shape[0]
    w = lambda n: p[n].reshape(p[n].shape[0], -1)

    def bn_relu(h, pre, key):
        _, _, sc, sh = _bn_train(h, p[pre + '.weight'], p[pre + '.bias'], eps)
        z = _r(h) * sc + sh
        return _r(torch.relu(z) if gates is None else z * (gates[key] > 0).float())

    def pool(h, key):
        h = h.view(G, gs, E)
        return h.max(1).values if gates is None else h.gather(1, gates[key].long().unsqueeze(1)).squeeze(1)

    f1 = bn_relu(x6 @ w('first_conv.0.weight').t(), 'first_conv.1', 'F1')      # hi/lo split: fp32-class product
    f2 = bn_relu(f1 @ _r(w('first_conv.3.weight')).t(), 'first_conv.4', 'F2')
    f = _r(f2 @ _r(w('first_conv.6.weight')).t() + p['first_conv.6.bias'])
    fg = pool(f, 'arg3')
    W4 = w('second_conv.0.weight')
    u = fg @ _r(W4[:, :E]).t()
    f4 = bn_relu(f @ _r(W4[:, E:]).t() + u.repeat_interleave(gs, 0), 'second_conv.1', 'F4')
    h5 = _r(f4 @ _r(w('second_conv.3.weight')).t())
    return pool(h5, 'arg5') + p['second_conv.3.bias']
