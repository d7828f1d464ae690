"""Which fp32 operation order do torch's Adam / RAdam use on this device?  (diagnostic for csrc/optim.hip)"""
import torch
torch.manual_seed(0)
n = 1 << 20
p0 = torch.randn(n, device='cuda'); g = torch.randn(n, device='cuda')
a = p0.clone().requires_grad_(True)
opt = torch.optim.RAdam([a], lr=1e-3, foreach=False)
f32 = lambda x: torch.tensor(x, dtype=torch.float32, device='cuda')
for step in range(1, 9):
    pprev = a.detach().clone()
    a.grad = g.clone() * step
    opt.step()
    m, v = opt.state[a]['exp_avg'], opt.state[a]['exp_avg_sq']
    b1, b2, eps, lr = 0.9, 0.999, 1e-8, 1e-3
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    rho_inf = 2 / (1 - b2) - 1
    rho_t = rho_inf - 2 * step * (b2 ** step) / bc2
    inv_f = f32(1.0) / f32(bc1)
    inv_d = f32(1.0 / bc1)
    for iname, inv in (("inv_f32", inv_f), ("inv_f64", inv_d)):
        bce = m * inv
        upd = bce * f32(lr)
        if rho_t > 5:
            rect = ((rho_t - 4) * (rho_t - 2) * rho_inf / ((rho_inf - 4) * (rho_inf - 2) * rho_t)) ** 0.5
            den = v.sqrt() + f32(eps)
            for aname, ad in (("rcp*s", (f32(1.0) / den) * f32(bc2 ** 0.5)), ("s/den", f32(bc2 ** 0.5) / den)):
                u = (upd * ad) * f32(rect)
                print(step, iname, aname, "mismatches:", int(((pprev - u) != a.detach()).sum()))
        else:
            print(step, iname, "unrect mismatches:", int(((pprev - upd) != a.detach()).sum()),
                  " m/bc1 variant:", int(((pprev - (m / f32(bc1)) * f32(lr)) != a.detach()).sum()))
