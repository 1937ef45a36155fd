"""Hyper-parameters of the RetuneGAN hot path: a plain module of attributes, imported as `hp` / `h` everywhere,
with the same names and defaults as the reference's retunegan/hparam.py (the API surface §8b of SURVEY.md keeps)."""
from sys import platform

# --- audio (kept in sync with the acoustic model's settings; retunegan/hparam.py:3-31)
sample_rate = 22050
n_fft = 2048
win_length = 1024
hop_length = 256
n_mel = 80
n_freq = 1025
preemphasis = 0.97
ref_level_db = 20
min_level_db = -100
max_abs_value = 4
trim_below_peak_db = 35
fmin = 125
fmax = 7600
rf0min = 'D2'
rf0max = 'D5'
c0min = 4.6309418394230306e-05
c0max = 0.3751049339771271
f0min = 73.25581359863281
f0max = 595.9459228515625
n_tone = 5 + 1
n_prds = 5 + 1
n_c0_bins = 32
n_f0_bins = None
n_f0_min = None
maxlen_text = 128
maxlen_spec = 1024

# --- vocoder audio (hparam.py:35-41)
segment_size = 8192
window_fn = 'hann'
mel_scale = 'slaney'
gl_iters = 4
gl_momentum = 0.7
gl_power = 1.2
ref_wav = 'y'

# --- generator (hparam.py:59-65)
generator_ver = 'RefineGAN_small'
split_cv = generator_ver.endswith('Split')
upsample_rates = [8, 8, 4]
upsample_kernel_sizes = [15, 15, 7]
upsample_initial_channel = 256
resblock_kernel_sizes = [3, 5, 7]
resblock_dilation_sizes = [[1, 2], [2, 6], [3, 12]]

# --- discriminators (hparam.py:70-83)
msd_layers = 3
mpd_periods = [3, 5, 7, 11]
multi_stft_params = [
    # (n_fft, win_length, hop_length); the STFT kernel takes n_fft = a power of two in 128 .. 4096 (RtgError otherwise)
    (2048, 1024, 240),
    (1024, 512, 120),
    (512, 256, 60),
]
phd_layers = len(multi_stft_params)
phd_input = 'stft'

# --- losses (hparam.py:86-91)
relative_gan_loss = False
strip_mirror_loss = False
dynamic_loss = True
envelope_loss = False
envelope_pool_k = 160
downsample_pool_k = 4

# --- misc (hparam.py:95-97)
debug = platform == 'win32'
randseed = 114514

# --- training (hparam.py:101-114)
num_workers = 1 if debug else 4
batch_size = 4 if debug else 16
learning_rate_d = 2e-4
learning_rate_g = 1.8e-4
d_train_times = 2
adam_b1 = 0.8
adam_b2 = 0.99
lr_decay = 0.999
w_loss_fm = 2
w_loss_mstft = 8
w_loss_env = 4
w_loss_dyn = 4
w_loss_sm = 0.01

# --- eval (hparam.py:118)
valid_limit = batch_size * 4

# --- extension (not in the reference): arithmetic of the convolutions on the MI355X path.  'fp32' = exact fp32 matrix
# cores (BASELINE configs[1]); 'bf16' = operands rounded to bf16 on the bf16 matrix cores, fp32 accumulation, fp32 tensors,
# fp32 losses and optimizer (BASELINE configs[2]); read when a model's weight bank is built.
compute_dtype = 'fp32'
# with compute_dtype 'bf16': the feature maps between the dense discriminator layers (and their gradients) live in HBM as
# bf16, stored activated — bf16(leaky_relu(x, 0.15)), what every consumer inside the stacks applies to them anyway; fp32
# accumulators, losses, weight norm and optimizer.  The fmaps the discriminators return are then such bf16 tensors
# (feature_loss reads them; rtg.ops.decode gives the fp32 feature map).  False: fp32 tensors in HBM, bf16 operands only.
bf16_maps = False
# resume schedule: False = the installed torch's ExponentialLR (2.x: the constructor leaves the loaded lr untouched);
# True = torch 1.8's (the reference's README.md:15): one more factor of lr_decay per resume.  See train.ExponentialLR.
legacy_resume_lr = False
