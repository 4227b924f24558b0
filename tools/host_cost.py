import os, sys, time
sys.path[:0] = ['/root/repo', '/root/repo/yolo2-pytorch_amd']
import torch, bench_data, train as y2train, utils
dev = torch.device('cuda:0')
for B, S in ((2, 64), (64, 416)):
    inf, anchors = bench_data.build_model(20, dev, 'darknet')
    inf.train()
    opt = utils.optim.SGD(inf.parameters(), 1e-3, momentum=0.9)
    data = {k: v.to(dev) for k, v in bench_data.labels(B, S, 20, seed=2).items()}
    data['tensor'] = bench_data.images(B, S, seed=11).to(dev)
    for _ in range(4):
        y2train.iterate(inf, opt, data, bench_data.HPARAM, 0.6, anchors)
    torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        t0 = time.perf_counter()
        y2train.iterate(inf, opt, data, bench_data.HPARAM, 0.6, anchors)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        ts.append(((t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
    print('B=%d S=%d host enqueue ms / total ms:' % (B, S), ' '.join('%.1f/%.1f' % t for t in ts))
