import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import warnings; warnings.filterwarnings("ignore")
from oracle import model_ref as M
import vqgan_training_amd as vq
print("cpu_count", os.cpu_count(), "torch threads", torch.get_num_threads(), flush=True)
for thr in (int(sys.argv[1]) if len(sys.argv) > 1 else torch.get_num_threads(),):
    torch.set_num_threads(thr)
    vae = vq.ae.VAE(256, 3, 128, 3, [1,2,4,4], 2, 16, False, False, False)
    lp = vq.utils.LPIPS(pretrained_path=None); disc = vq.utils.PatchDiscriminator()
    st = M.RefState(vae.state_dict(), lp.state_dict(), disc.state_dict())
    kw = dict(do_ganloss=True, disc_type="hinge", learning_rate_vae=1e-5, vae_ch=128, max_steps=1000)
    for res in (64, 128, 256):
        x = torch.rand(1,3,res,res)*2-1
        t0=time.time(); M.train_step_ref(st, x, **kw); print("threads", thr, "res", res, "%.2fs"%(time.time()-t0), flush=True)
