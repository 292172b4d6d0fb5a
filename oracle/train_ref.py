"""ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of one GIF training iteration (train.py:82-250) on top of
oracle/stylegan2_ref.py: same losses, R1, Adam hyper-parameters and EMA as the reference loop.  Used by tests to
check gif_amd.train_step and by bench.py's `cpu_baseline` leg (kind "port") — never by the product path."""
import torch
import torch.nn.functional as F

from . import stylegan2_ref as R


def _leaves(sd, trainable):
    return {k: (v.clone().requires_grad_(True) if trainable(k) else v.clone()) for k, v in sd.items()}


class RefTrainer:
    def __init__(self, g_sd, d_sd, res_step=6, size=256, r1_every=16, lr=0.002):
        self.g = _leaves(g_sd, lambda k: not k.endswith('kernel') and 'embd_weight' not in k)
        self.d = _leaves(d_sd, lambda k: not k.endswith('kernel'))
        self.g_ema = {k: v.detach().clone() for k, v in self.g.items()}
        self.res_step, self.size, self.r1_every = res_step, size, r1_every
        gr, dr = 4 / 5, 16 / 17  # train.py:364-381
        self.g_params = [v for v in self.g.values() if v.requires_grad]
        self.d_params = [v for v in self.d.values() if v.requires_grad]
        self.g_opt = torch.optim.Adam(self.g_params, lr=lr * gr, betas=(0.0, 0.99 ** gr))
        self.d_opt = torch.optim.Adam(self.d_params, lr=lr * dr, betas=(0.0, 0.99 ** dr))

    def texture_interp_loss(self, tex):
        """train.py:222-238 + loss_functions/losses.py:162-243 on pre-rendered inputs.  tex: dict with
          gen_in [N,6,R,R] (rendered condition of the interpolated FLAME batch), identities [N] int64 (the one fixed identity),
          verts / normals [N,V,3] and cam [N,3] (FlameTextureSpace.forward's mesh, stg2_generator.py:355-376), texture_data (the
          dict of :348-353), face_mask [1,1,h,w], pairs [P,2] (the np.random.choice draw of :166-167)."""
        from . import texture_loss_ref as TL
        from . import texture_ref as TR
        img = R.generator_forward(self.g, tex["gen_in"], self.res_step, tex["identities"])
        textures, masks = TR.compute_texture_map(tex["texture_data"], img, tex["verts"], tex["normals"], tex["cam"])
        return TL.texture_pairs_loss(tex["face_mask"], textures, masks, tex["pairs"])

    def step(self, i, real, cond, idx, tex=None, adaptive_interp_loss=False):
        # ---- D step, train.py:82-178
        self.d_opt.zero_grad(set_to_none=True)
        real = real.detach().requires_grad_(True)
        rs = R.discriminator_forward(self.d, real, cond, self.size)
        d_loss = F.softplus(-rs).mean()
        if self.r1_every and (i + 1) % self.r1_every == 0:
            d_loss = d_loss + R.grad_penalty_loss([real], rs).mean()
        with torch.no_grad():
            fake = R.generator_forward(self.g, cond, self.res_step, idx)
        fs = R.discriminator_forward(self.d, fake, cond, self.size)
        d_loss = d_loss + F.softplus(fs).mean()
        d_loss.backward()
        self.d_opt.step()
        # ---- G step, train.py:189-252
        self.g_opt.zero_grad(set_to_none=True)
        for p in self.d_params:
            p.requires_grad_(False)
        fake = R.generator_forward(self.g, cond, self.res_step, idx)
        g_loss = F.softplus(-R.discriminator_forward(self.d, fake, cond, self.size)).mean()
        if tex is not None:
            interp_loss = self.texture_interp_loss(tex)
            if adaptive_interp_loss:  # train.py:236-237
                interp_loss = interp_loss * (0.25 * g_loss.detach() / interp_loss.detach())
            g_loss = g_loss + interp_loss
        g_loss.backward()
        for p in self.d_params:
            p.requires_grad_(True)
        self.g_opt.step()
        decay = 0.5 ** (32 / (10 * 1000))
        with torch.no_grad():  # generic_utils.accumulate :63-76
            for k, v in self.g.items():
                if v.requires_grad:
                    self.g_ema[k].mul_(decay).add_(v.detach(), alpha=1 - decay)
        return d_loss.detach(), g_loss.detach()
