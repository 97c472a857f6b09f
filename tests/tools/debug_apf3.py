"""Debug aid: BatchNorm running statistics of the APF encoder after the fixture test's call sequence."""
import sys, torch
sys.path.insert(0, '.')
import numpy as np
import p4p_amd
from p4p_amd import apf as PA, functional_apf as FA, ops
from oracle import apf_oracle as A

z = np.load('tests/golden/apf_encoder.npz')
for tag, ec in (('ec128', 128), ('ec768', 768)):
    for seq in ('train', 'eval+train', 'eval+train+bwd'):
        m = PA.Encoder(ec, 6); m.load_state_dict(A.seeded_state_dict(A.Encoder(ec, 6), seed=900 + ec)); m = m.to('cuda')
        x = torch.from_numpy(z[f'{tag}_x']).cuda()
        if 'eval' in seq:
            m.eval()
            with torch.no_grad():
                m(x)
        m.train()
        tok = m(x)
        if 'bwd' in seq:
            (tok * torch.from_numpy(z[f'{tag}_train_gradw']).cuda()).sum().backward()
        for bn, key in ((m.first_conv[1], 'bn1_running'), (m.second_conv[1], 'sbn_running')):
            run = np.stack([bn.running_mean.cpu().numpy(), bn.running_var.cpu().numpy()])
            d = np.abs(run - z[f'{tag}_{key}'])
            print(tag, seq, key, 'mean err', d[0].max(), 'var err', d[1].max(), 'var rel', (d[1] / z[f'{tag}_{key}'][1]).max())
