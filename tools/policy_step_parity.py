"""One train step of a toy model (ch=32, 1,2) in a precision policy against the CPU oracle: python tools/policy_step_parity.py <policy> [res].
GPU when there is one, else the host emulator (tests/emu)."""
import sys, warnings, time
warnings.simplefilter("ignore")
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import vqgan_training_amd as vq
from vqgan_training_amd import ops
from oracle import model_ref as M, weights as W
from conftest import EMU_LIB, HIP_LIB
policy = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
gpu = torch.cuda.is_available()
lib = vq._lib.VqLibrary(HIP_LIB if gpu else EMU_LIB)
vq._lib._set_library_for_tests(lib)
dev = torch.device("cuda:0" if gpu else "cpu")
def rel(a, b):
    a, b = torch.as_tensor(a).detach().float().cpu(), torch.as_tensor(b).detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()
res, ch, mult = (int(sys.argv[2]) if len(sys.argv) > 2 else 16), 32, [1, 2]
vae = vq.ae.VAE(res, 3, ch, 3, list(mult), 1, 4, False, False, False)
vae.load_state_dict(W.randomize_state_dict(vae.state_dict(), 1))
lp = vq.utils.LPIPS(pretrained_path=None)
lp.load_state_dict(W.randomize_state_dict(lp.state_dict(), 2, relu_net=True))
disc = vq.utils.PatchDiscriminator()
disc.load_state_dict(W.randomize_state_dict(disc.state_dict(), 4, relu_net=True))
st = M.RefState(vae.state_dict(), lp.state_dict(), disc.state_dict())
vae, lp, disc = vae.to(dev), lp.to(dev).eval(), disc.to(dev)
vq.vae_trainer.apply_precision_policy(policy, vae, lp, disc)
grads = {}
kw = dict(do_ganloss=True, disc_type="hinge", learning_rate_vae=2e-3, vae_ch=ch, max_steps=10, warmup_steps=0)
step = vq.vae_trainer.VAETrainStep(vae, lp, disc, on_backward=lambda s: grads.update({n: p.grad.detach().clone() for n, p in vae.named_parameters()}) if not grads else None, **kw)
x = W.image_batch(2, res, seed=8)
t0 = time.time()
print("calibration:", step.calibrate_grad_scales(x.to(dev)))
o = step(x.to(dev))
print("step time", time.time() - t0)
r = M.train_step_ref(st, x, **kw)
for k in ("overall_vae_loss", "perceptual_loss", "vae_loss", "d_loss", "g_gan_loss"):
    print(k, float(o[k]), float(r[k]), f"{rel(o[k], r[k]):.3e}")
print("recon", f"{rel(o['reconstructed'], r['reconstructed']):.3e}", "z", f"{rel(o['z'], r['z']):.3e}")
num = sum(((grads[k].cpu() - v) ** 2).sum().item() for k, v in r["grads"].items())
den = sum((v ** 2).sum().item() for v in r["grads"].values())
print("grad_l2", (num / den) ** 0.5)
print(step.poll_range_events())
