import numpy as np, torch
x = np.linspace(-708.45, -707.9, 23)
g = torch.exp(torch.from_numpy(x).cuda()).cpu().numpy()
c = np.exp(x)
print("tiny", np.finfo(float).tiny)
for a, b, d in zip(x, c, g):
    print("%.4f numpy %.6e device %.6e rel %.2e" % (a, b, d, abs(d - b) / b))
