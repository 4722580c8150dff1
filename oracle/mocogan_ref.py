"""ORACLE (test infrastructure only) -- CPU restatement of the MoCoGAN + Neural-ODE hot path on stock torch.nn.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(gan-ode_amd/) never does.

Each builder cites the reference lines it follows; layer order, hyper-parameters, RNG call order, reshapes and
state_dict keys follow the reference exactly so that (a) weights can be exchanged by state_dict and (b) the
goldens generated from the reference's own classes (oracle/make_goldens.py -> tests/golden/) pin this file in
tests/test_oracle_golden.py.  The integrator comes from oracle/ode_ref.py (third-party arithmetic, see its
header: "parity unpinned" for that piece).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import ode_ref


# --------------------------------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------------------------------
class OdeRhs(nn.Module):
    """f(t, x) = W2 tanh(W1 x + b1) + b2, autonomous.  /root/reference/models/mocogan_ode.py:6-17."""

    def __init__(self, dim, dim_hidden):
        super().__init__()
        self.fn = nn.Sequential(nn.Linear(dim, dim_hidden), nn.Tanh(), nn.Linear(dim_hidden, dim))

    def forward(self, t, x):
        return self.fn(x)


def _prenet(dim):
    # /root/reference/models/mocogan_ode.py:123-129 (also :29-35)
    return nn.Sequential(nn.Linear(dim, 64), nn.LeakyReLU(0.2), nn.Linear(64, dim), nn.LeakyReLU(0.2))


def _decoder(dim_z, ngf, n_channels, mnist):
    """Frame decoder.  UCF: /root/reference/models/mocogan.py:200-215; MNIST 28x28: mocogan_ode.py:66-84."""
    seq = []
    widths = [dim_z, ngf * 8, ngf * 4, ngf * 2, ngf]
    for i in range(4):
        stride, pad = (1, 0) if i == 0 else (2, 1)
        seq += [nn.ConvTranspose2d(widths[i], widths[i + 1], 4, stride, pad, bias=False),
                nn.BatchNorm2d(widths[i + 1]), nn.ReLU(True)]
    if mnist:
        seq.append(nn.ConvTranspose2d(ngf, n_channels, kernel_size=1, stride=1, padding=2, bias=False))
    else:
        seq.append(nn.ConvTranspose2d(ngf, n_channels, 4, 2, 1, bias=False))
    seq.append(nn.Tanh())
    return nn.Sequential(*seq)


class Generator(nn.Module):
    """VideoGeneratorMNISTODE (mnist=True; mocogan_ode.py:114-148) / ODE VideoGenerator (mnist=False;
    mocogan_ode.py:20-54) on top of the MoCoGAN base generator (mocogan.py:185-301)."""

    def __init__(self, n_channels, dim_z_content, dim_z_category, dim_z_motion, video_length, dim_hidden=None,
                 ngf=64, mnist=True, ode_method="rk4", ode_options=None):
        super().__init__()
        assert dim_z_category == 0, "the stage-3 scripts never use categories (mnist_moco_ode.py:78)"
        self.n_channels, self.dim_z_content, self.dim_z_category = n_channels, dim_z_content, dim_z_category
        self.dim_z_motion, self.video_length = dim_z_motion, video_length
        self.recurrent = nn.GRUCell(dim_z_motion, dim_z_motion)  # dead weight kept for state_dict parity
        self.main = _decoder(dim_z_motion + dim_z_category + dim_z_content, ngf, n_channels, mnist)
        self.ode_fn = OdeRhs(dim_z_motion, dim_hidden if dim_hidden else dim_z_motion)
        self.linear = _prenet(dim_z_motion)
        self.ode_method, self.ode_options = ode_method, ode_options

    def _dtype(self):
        """fp32 as the reference; a .double() copy of the module gives an fp64 yardstick fed by the SAME fp32 draws."""
        return self.ode_fn.fn[0].weight.dtype

    # -- latent samplers; RNG call order is part of the contract (mocogan.py:249-269, mocogan_ode.py:133-148)
    def sample_z_content(self, n, video_len=None):
        T = video_len or self.video_length
        c = np.random.normal(0, 1, (n, self.dim_z_content)).astype(np.float32)
        return torch.from_numpy(np.repeat(c, T, axis=0)).to(self._dtype())

    def sample_z_m(self, n, video_len=None):
        T = video_len or self.video_length
        x = self.linear(torch.randn(n, self.dim_z_motion).to(self._dtype()))
        sol = ode_ref.odeint_adjoint(self.ode_fn, x, torch.linspace(0, 1, T).float().to(self._dtype()),
                                     method=self.ode_method,
                                     options=self.ode_options)
        return sol.transpose(0, 1).reshape(-1, self.dim_z_motion)

    def sample_z_video(self, n, video_len=None):
        zc = self.sample_z_content(n, video_len)
        zm = self.sample_z_m(n, video_len)
        return torch.cat([zc, zm], dim=1), np.zeros(n)

    def sample_videos(self, n, video_len=None):
        T = video_len or self.video_length
        z, labels = self.sample_z_video(n, T)
        h = self.main(z.view(z.size(0), z.size(1), 1, 1))
        h = h.view(h.size(0) // T, T, self.n_channels, h.size(3), h.size(3)).permute(0, 2, 1, 3, 4)
        return h, torch.from_numpy(labels)

    def sample_images(self, n):
        z, _ = self.sample_z_video(n * self.video_length * 2)
        j = np.sort(np.random.choice(z.size(0), n, replace=False)).astype(np.int64)
        z = z[j]
        return self.main(z.view(z.size(0), z.size(1), 1, 1)), None


class GeneratorOdeRnn(Generator):
    """VideoGeneratorMNISTODERNN (/root/reference/models/mocogan_ode_rnn.py:21-53; un-importable as shipped because
    of `on_dev`, restated here): per frame h' = odeint_adjoint(ode_fn, h, [0, 1])[-1] with torchdiffeq's DEFAULT
    solver (dopri5, rtol 1e-7, atol 1e-9), then h = GRUCell(e_t, h').  The pre-net `linear` exists but is unused.
    Noise comes from the legacy FloatTensor(...).normal_() calls of models/mocogan.py:297-301 (global torch CPU
    generator); parity for this class is unpinned twice over (no fixture, torchdiffeq absent)."""

    def _normal(self, n):
        return torch.FloatTensor(n, self.dim_z_motion).normal_().to(self._dtype())

    def sample_z_m(self, n, video_len=None):
        T = video_len or self.video_length
        h = [self._normal(n)]
        t01 = torch.tensor([0, 1]).float().to(self._dtype())
        for _ in range(T):
            e = self._normal(n)
            hp = ode_ref.odeint_adjoint(self.ode_fn, h[-1], t01)[-1]
            h.append(self.recurrent(e, hp))
        return torch.cat([k.view(-1, 1, self.dim_z_motion) for k in h[1:]], dim=1).view(-1, self.dim_z_motion)


def build_mnist_odernn(ngf=64, ndf=64):
    """mnist_moco_ode_rnn.py:75-78 (same discriminators, ODE-RNN generator)."""
    return (GeneratorOdeRnn(1, 50, 0, 16, 16, ngf=ngf, mnist=True), VideoDisc(1, ksize=2, ndf=ndf),
            PatchImageDisc(1, ndf=ndf))


def _noise_slot():
    return nn.Identity()  # Noise(use_noise=False) is the identity (mocogan.py:20-29); keeps Sequential indices


class VideoDisc(nn.Module):
    """VideoDiscriminator, /root/reference/models/mocogan.py:129-164."""

    def __init__(self, n_channels, n_output_neurons=1, ndf=64, ksize=4):
        super().__init__()
        st, pd = (1, 2, 2), (0, 1, 1)
        w = [n_channels, ndf, ndf * 2, ndf * 4, ndf * 8]
        seq = [_noise_slot(), nn.Conv3d(w[0], w[1], ksize, stride=st, padding=pd, bias=False),
               nn.LeakyReLU(0.2, inplace=True)]
        for i in range(1, 4):
            seq += [_noise_slot(), nn.Conv3d(w[i], w[i + 1], ksize, stride=st, padding=pd, bias=False),
                    nn.BatchNorm3d(w[i + 1]), nn.LeakyReLU(0.2, inplace=True)]
        seq.append(nn.Conv3d(w[4], n_output_neurons, ksize, 1, 0, bias=False))
        self.main = nn.Sequential(*seq)

    def forward(self, x):
        return self.main(x).squeeze(), None


class PatchImageDisc(nn.Module):
    """PatchImageDiscriminator, /root/reference/models/mocogan.py:66-93."""

    def __init__(self, n_channels, ndf=64):
        super().__init__()
        seq = [_noise_slot(), nn.Conv2d(n_channels, ndf, 4, 2, 1, bias=False), nn.LeakyReLU(0.2, inplace=True)]
        for cin, cout in ((ndf, ndf * 2), (ndf * 2, ndf * 4)):
            seq += [_noise_slot(), nn.Conv2d(cin, cout, 4, 2, 1, bias=False), nn.BatchNorm2d(cout),
                    nn.LeakyReLU(0.2, inplace=True)]
        seq += [_noise_slot(), nn.Conv2d(ndf * 4, 1, 4, 2, 1, bias=False)]
        self.main = nn.Sequential(*seq)

    def forward(self, x):
        return self.main(x).squeeze(), None


# --------------------------------------------------------------------------------------------------------------
# the training iteration (mnist_moco_ode.py:113-163 == ucf_moco_ode.py:115-165)
# --------------------------------------------------------------------------------------------------------------
def make_optimizers(gen, dis_vid, dis_img):
    """mnist_moco_ode.py:86-88 -- Adam with L2-coupled weight decay."""
    mk = lambda m: torch.optim.Adam(m.parameters(), lr=2e-4, betas=(0.5, 0.999), weight_decay=1e-5)  # noqa: E731
    return mk(gen), mk(dis_vid), mk(dis_img)


def d_image_step(gen, dis_img, opt, real_img, bce, batch):
    opt.zero_grad()
    pr, _ = dis_img(real_img)
    with torch.no_grad():
        fake, _ = gen.sample_images(batch)
    pf, _ = dis_img(fake)
    loss = bce(pr, torch.ones_like(pr)) + bce(pf, torch.zeros_like(pf))
    loss.backward()
    opt.step()
    return loss


def d_video_step(gen, dis_vid, opt, real_vid, bce, batch):
    opt.zero_grad()
    pr, _ = dis_vid(real_vid.transpose(1, 2))
    with torch.no_grad():
        fake, _ = gen.sample_videos(batch)
    pf, _ = dis_vid(fake)
    loss = bce(pr, torch.ones_like(pr)) + bce(pf, torch.zeros_like(pf))
    loss.backward()
    opt.step()
    return loss


def g_step(gen, dis_vid, dis_img, opt, bce, batch):
    opt.zero_grad()
    fake_vid, _ = gen.sample_videos(batch)
    fake_img, _ = gen.sample_images(batch)
    pv, _ = dis_vid(fake_vid)
    pi, _ = dis_img(fake_img)
    loss = bce(pv, torch.ones_like(pv)) + bce(pi, torch.ones_like(pi))
    loss.backward()
    opt.step()
    return loss


def train_step(gen, dis_vid, dis_img, opts, real_imgs, real_vids, d_iters=2):
    """One outer iteration.  real_imgs / real_vids: sequences of d_iters batches ([B,C,H,W] / [B,T,C,H,W]).
    Returns (dis_img_loss, dis_vid_loss, gen_loss) of the LAST inner pass, as the reference prints them."""
    gen_opt, vid_opt, img_opt = opts
    bce = nn.BCEWithLogitsLoss()
    batch = real_imgs[0].shape[0]
    for i in range(d_iters):
        li = d_image_step(gen, dis_img, img_opt, real_imgs[i], bce, batch)
        lv = d_video_step(gen, dis_vid, vid_opt, real_vids[i], bce, batch)
    lg = g_step(gen, dis_vid, dis_img, gen_opt, bce, batch)
    return li.detach(), lv.detach(), lg.detach()


def train_step_dp(gen, dis_vid, dis_img, opts, real_img_shards, real_vid_shards, d_iters=2):
    """The data-parallel golden of SURVEY 8(e): S replicas, each running the reference's forward/backward on its own
    shard with its OWN BatchNorm batch statistics, gradients AVERAGED over the replicas before every optimiser step
    (what one all-reduce(sum)/S per step computes).  real_img_shards[i][s] / real_vid_shards[i][s]: shard s of inner
    pass i.  The replicas are evaluated one after the other on ONE set of weights (they are identical in DP), drawing
    their noise from the global generators in shard order; .grad accumulates over the shards (autograd adds) and is
    divided by S.  Returns the three losses of the last inner pass averaged over shards.  (BatchNorm running statistics
    see S updates per step here, a replica sees one: they do not enter train-mode arithmetic.)"""
    gen_opt, vid_opt, img_opt = opts
    bce = nn.BCEWithLogitsLoss()
    S = len(real_img_shards[0])
    batch = real_img_shards[0][0].shape[0]

    def averaged(opt, model, shard_losses):
        opt.zero_grad()
        tot = 0.0
        for fn in shard_losses:
            loss = fn()
            loss.backward()
            tot = tot + loss.detach()
        for p in model.parameters():
            if p.grad is not None:
                p.grad.div_(S)
        opt.step()
        return tot / S

    def img_loss(x):
        def fn():
            pr, _ = dis_img(x)
            with torch.no_grad():
                fake, _ = gen.sample_images(batch)
            pf, _ = dis_img(fake)
            return bce(pr, torch.ones_like(pr)) + bce(pf, torch.zeros_like(pf))
        return fn

    def vid_loss(x):
        def fn():
            pr, _ = dis_vid(x.transpose(1, 2))
            with torch.no_grad():
                fake, _ = gen.sample_videos(batch)
            pf, _ = dis_vid(fake)
            return bce(pr, torch.ones_like(pr)) + bce(pf, torch.zeros_like(pf))
        return fn

    def gen_loss():
        fake_vid, _ = gen.sample_videos(batch)
        fake_img, _ = gen.sample_images(batch)
        pv, _ = dis_vid(fake_vid)
        pi, _ = dis_img(fake_img)
        return bce(pv, torch.ones_like(pv)) + bce(pi, torch.ones_like(pi))

    for i in range(d_iters):
        li = averaged(img_opt, dis_img, [img_loss(x) for x in real_img_shards[i]])
        lv = averaged(vid_opt, dis_vid, [vid_loss(x) for x in real_vid_shards[i]])
    # (the generator step's backward also fills the discriminators' .grad; the next zero_grad discards it)
    lg = averaged(gen_opt, gen, [gen_loss] * S)
    return li, lv, lg


def build_mnist(ngf=64, ndf=64):
    """mnist_moco_ode.py:75-78."""
    return (Generator(1, 50, 0, 16, 16, ngf=ngf, mnist=True), VideoDisc(1, ksize=2, ndf=ndf),
            PatchImageDisc(1, ndf=ndf))


def build_ucf(ngf=64, ndf=64):
    """ucf_moco_ode.py:77-80 with dim_hidden=16 -- the shipped ctor call omits dim_hidden and raises TypeError
    (SURVEY.md section 0.1); 16 is the minimal repair and is recorded as a deviation."""
    return (Generator(3, 50, 0, 16, 16, dim_hidden=16, ngf=ngf, mnist=False), VideoDisc(3, ndf=ndf),
            PatchImageDisc(3, ndf=ndf))
