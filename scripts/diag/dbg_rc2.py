import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from test_mlp_rc_gpu import _fine_setup
from test_mlp_rc2_gpu import _forward2
dev = torch.device('cuda:0')
for M in (1000, 992, 1024, 2000):
    X0, Z, Ws, bs, relu = _fine_setup(M, dev, seed=M)
    outs, bits = _forward2(M, X0, Z, Ws, bs, relu, dev)
    torch.cuda.synchronize()
    for i in range(7):
        bad = (~torch.isfinite(outs[i][:, :256])).any(1).nonzero().flatten().tolist()
        if bad:
            cols = (~torch.isfinite(outs[i][bad[0], :256])).nonzero().flatten().tolist()
            print(M, 'layer', i, 'bad rows', len(bad), bad[:12], '... cols of first', cols[:6], len(cols))
print('done')
