"""Debug: what differs after a failed whole-step capture (labels? logits?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_parity_gpu as T
from bias_gan_amd.utils import losses, parsing_helpers as ph
from bias_gan_amd.gpsro_train.train_gan import GANTrainer

c, h, w, n = 4, 64, 64, 2
def run(flag):
    os.environ["BGAMD_STEP_GRAPH"] = flag
    G, _ = T.build_generator(c, 41, T.F32)
    D, _ = T.build_discriminator(c, h, w, 42, T.F32)
    G.train(), D.train()
    crit = losses.GANLoss("ModifiedMinMax", n, torch.device("cuda"))
    orig = crit.d_loss
    rec = []
    def d_loss(lr_, lf_, labels=None):
        if labels is None:
            labels = crit.draw_labels()
        if not torch.cuda.is_current_stream_capturing():
            rec.append(("labels", [float(x) for x in labels[0].flatten()], [float(x) for x in labels[1].flatten()], labels[2],
                        [float(x) for x in lr_.flatten()], [float(x) for x in lf_.flatten()]))
        return orig(lr_, lf_, labels)
    crit.d_loss = d_loss
    g_opt = ph.get_optimizer(G.parameters(), "Adam", 0.0, 1e-8, 0.0)
    d_opt = ph.get_optimizer(D.parameters(), "Adam", 0.0, 1e-8, 0.0)
    tr = GANTrainer(G, D, g_opt, d_opt, crit, T._SyncingL1())
    out = []
    for s_ in range(5):
        torch.manual_seed(300 + s_)
        x, y = (t.to("cuda") for t in T.orc.synthetic_fields(n, c, h, w, 700 + s_))
        d, g = tr.step(x, y)
        out.append((float(d), float(g)))
    return out, rec
import warnings
warnings.simplefilter("ignore")
e, re_ = run("0")
g, rg = run("1")
for i in range(5):
    print(i, e[i], g[i])
for a, b in zip(re_, rg):
    print("E", a[1:]); print("G", b[1:])
