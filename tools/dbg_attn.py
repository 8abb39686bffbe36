import sys, torch, ctypes
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from mmvid_amd import ops, _lib
torch.manual_seed(0)
B, L, H = 1, 64, 12
E = H * 64
qkv = torch.randn(B * L, 3 * E, device='cuda').bfloat16()
q, k, v = qkv.float().view(B, L, 3, H, 64).permute(2, 0, 3, 1, 4)
ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).permute(0, 2, 1, 3).reshape(B * L, E)
vts = [ops.head_transpose(qkv, 2 * E, B, L, H) for _ in range(3)]
vt_ref = torch.zeros(B, H, 64, 64, device='cuda', dtype=torch.bfloat16)
vt_ref[:, :, :, :L] = qkv[:, 2 * E:].view(B, L, H, 64).permute(0, 2, 3, 1)
print('head_transpose exact:', [torch.equal(x, vt_ref) for x in vts])
VT = vts[0]
res = []
for it in range(40):
    out = torch.full((B * L, E), float('nan'), device='cuda', dtype=torch.bfloat16)
    lse = torch.empty(B, H, L, device='cuda')
    _lib.call('mmvid_attention_fwd', ops._p(qkv), 3 * E, ops._p(VT), B, L, 64, H, E, 0.125, 0, -1, 0, -1, 0, ops._p(out), E, ops._p(lse), ops._stream())
    torch.cuda.synchronize()
    err = (out.float() - ref).abs().view(L, H, 64).amax(-1)
    res.append(out)
    print(it, 'maxerr', err.max().item(), 'bad heads', sorted(set(h for _, h in (err > 0.01).nonzero().tolist())), 'nan', torch.isnan(out.float()).sum().item())
    if (err > 0.01).any():
        r, h = (err > 0.01).nonzero()[0].tolist()
        print('   first bad row', r, 'head', h, 'got', out[r, h*64:h*64+8].float().tolist(), 'ref', [round(x,3) for x in ref[r, h*64:h*64+8].tolist()])
