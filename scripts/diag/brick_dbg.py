import sys, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import test_brick_adam_gpu as T
dev = torch.device('cuda:0')
a, ca, ua = T._train(dev, 4, brick=True)
print('used', ua, 'clean', ca['k0_grad']['clean'])
b, cb, ub = T._train(dev, 4, brick=False)
print('used_b', ub, 'k0_grad' in cb)
for pa, pb in zip(a, b):
    print(tuple(pa.shape), float((pa - pb).norm() / pb.norm().clamp_min(1e-30)))
