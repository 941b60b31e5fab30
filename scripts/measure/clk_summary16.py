import sys, numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16, 16)
raw = raw[raw[:, :, :7].sum(axis=(1, 2)) > 0]
raw[:, :, 6] &= np.uint64((1 << 40) - 1); raw[:, :, 7] &= np.uint64((1 << 40) - 1)
a = raw.astype(float) / float(sys.argv[2])
names = ['phase1', 'prefetch+coltot', 'scan+write (+B4..loop top in band)', 'barrier(SAT) (+top..wait)', 'gap+lists+wait', 'batches', 'barrier(end)', 'nbatch',
         'leave loop+descriptor', 'first loads issued', 'fold', 'flush', 'column weights+barrier', '-', '-', '-']
tot = a[:, :, :7].sum(axis=2) + a[:, :, 8:13].sum(axis=2)
print('ticks per wave per chromosome: %.0f' % tot.mean())
np.set_printoptions(linewidth=220, suppress=True)
for i, nm in enumerate(names):
    if nm == '-': continue
    print('%-36s %8.0f (%4.1f%%)' % (nm, a[:, :, i].mean(), 100 * a[:, :, i].mean() / tot.mean()), a[:, :, i].mean(axis=0).round(0)[[0, 1, 8, 15]])
