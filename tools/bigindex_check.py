import torch, time
dev="cuda"
V, W = 134217728, 23
vol = torch.empty((V, W), dtype=torch.int32, device=dev)
# vol[v, w] = (v * 31 + w * 7) mod 2^31  computed in chunks
step = 1 << 24
wv = (torch.arange(W, device=dev, dtype=torch.int64) * 7).view(1, W)
for s in range(0, V, step):
    v = torch.arange(s, min(V, s + step), device=dev, dtype=torch.int64).view(-1, 1)
    vol[s:s + v.shape[0]] = ((v * 31 + wv) & 0x7FFFFFFF).to(torch.int32)
g = torch.Generator(device=dev); g.manual_seed(1)
n = 126000000
lin = torch.randint(0, V, (n,), device=dev, dtype=torch.int64, generator=g)
out = vol[lin]
bad = 0
for s in range(0, n, step):
    l = lin[s:s + step].view(-1, 1)
    want = ((l * 31 + wv) & 0x7FFFFFFF).to(torch.int32)
    bad += int((out[s:s + step] != want).sum().item())
print("advanced indexing mismatches:", bad, "of", n * W)
out2 = torch.index_select(vol, 0, lin[:20000000])
l = lin[:20000000].view(-1, 1)
print("index_select mismatches:", int((out2 != ((l * 31 + wv) & 0x7FFFFFFF).to(torch.int32)).sum().item()))
