"""Peak host memory and time of one oracle iteration (sizing the full-size tests). usage: oracle_mem.py C H W N"""
import os, resource, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import gan_oracle as orc
c, h, w, n = (int(a) for a in sys.argv[1:5])
torch.set_num_threads(16)
gspec, dspec = orc.generator_spec(c, c, 0, "batch"), orc.discriminator_spec(c, h, w, "batch")
st = orc.GANStep(orc.fill_state(gspec, 1), orc.fill_state(dspec, 2), orc.trainable_keys(gspec), orc.trainable_keys(dspec), "batch", "ModifiedMinMax")
x, y = orc.synthetic_fields(n, c, h, w, 333)
torch.manual_seed(11)
t0 = time.time(); d, g = st.step(x, y, labels=orc.draw_d_labels(n))
print(f"{c}x{h}x{w} N={n}: {time.time() - t0:.1f} s, peak RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20:.1f} GiB, d {d:.5f} g {g:.5f}")
