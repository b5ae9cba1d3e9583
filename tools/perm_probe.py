import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, frankenstein_amd as fa
fa.set_compute_dtype("bf16")
m, cfg = bench.cfg2_model("bf16"); bench.init_weights(m); m.cuda()
g = torch.Generator(device="cuda").manual_seed(3)
x = torch.randn(32, 600, 256, device="cuda", generator=g); y = torch.randn(32, 32, 128, device="cuda", generator=g)
perm = torch.randperm(32, device="cuda", generator=g)
with torch.no_grad():
    _, p1 = m(x, y); _, p2 = m(x, y)
    print("run-to-run identical:", torch.equal(p1, p2))
    e1 = m.encoder(x); e2 = m.encoder(x[perm].contiguous())
    print("encoder perm-equivariant:", torch.equal(e2, e1[perm]), float((e2.float() - e1[perm].float()).abs().max()))
    # layer by layer
    tr = m.encoder.transformer
    from frankenstein_amd.models.brainformer import _prep
    def run(xx, upto):
        h = m.encoder.embed(xx) if hasattr(m.encoder, "embed") else None
        return h
    hs = []
    def hook(mod, inp, out): hs.append(out.detach().clone())
    hooks = [blk.register_forward_hook(hook) for blk in tr.h]
    hs.clear(); m.encoder(x); a = list(hs); hs.clear(); m.encoder(x[perm].contiguous()); b = list(hs)
    for i, (u, v) in enumerate(zip(a, b)):
        print("block", i, "equal:", torch.equal(v, u[perm]), float((v.float() - u[perm].float()).abs().max()))
    for hk in hooks: hk.remove()
