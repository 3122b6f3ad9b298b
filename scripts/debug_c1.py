import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import cycle_diffusion_amd as cda
import golden_util as gu
from cycle_diffusion_amd import _ffi, schedule
from oracle import nets, samplers
eng = cda.Engine("cuda:0")
fx = gu.load("c1_toy_ddpm")
net = eng.create_net(cda.ho_ddpm_desc(32, 32, (1, 2, 2), 1, (16,)))
sd = gu.weights(fx); eng.load_state_dict(net, sd)
steps, eta = 50, 0.1
img = torch.rand((1, 3, 32, 32), generator=torch.Generator().manual_seed(11)); x0 = (img - 0.5) * 2
enc_noise, last = gu.pixel_noise(int(fx["noise_seed"]), x0.shape, steps)
sch = schedule.PixelSchedule(steps, steps, sample_type="ddim", eta=eta)
netf = lambda x, t: nets.ho_unet(sd, gu.TOY_HO_CFG, x, t)
with torch.no_grad():
    zo = torch.stack(samplers.pixel_encode(netf, x0, samplers.pixel_betas(), steps, steps, eta, enc_noise), 1)
ze = eng.dpm_encode(net, sch.kind, x0.cuda(), sch.coef_encode(), noise=torch.stack(enc_noise, 0).cuda(), last_uses_x0=False).cpu()
d = (ze - zo).flatten(2).abs().max(dim=2).values[0]
print("z slot maxabs diff:", [round(float(v), 4) for v in d])
print("z slot norms oracle:", [round(float(v), 2) for v in zo.flatten(2).norm(dim=2)[0][:6]])
for name, z in (("oracle z", zo), ("engine z", ze)):
    x = eng.ddim_decode(net, sch.kind, z.cuda(), sch.coef_decode(), n_eps=steps - 1, noise_tail=last[None].cuda()).cpu()
    with torch.no_grad():
        xo = samplers.pixel_decode(netf, z, samplers.pixel_betas(), steps, steps, eta, last)
    print(name, "engine-decode psnr vs img", gu.psnr((x + 1) / 2, img), "oracle-decode psnr vs img", gu.psnr((xo + 1) / 2, img),
          "engine vs oracle decode", gu.psnr((x + 1) / 2, (xo + 1) / 2))
cd = sch.coef_decode()
print(cd[:3]); print(cd[-2:])
