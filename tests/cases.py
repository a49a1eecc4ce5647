"""Synthetic stream shapes shared by the CPU and GPU parity tests: every subframe type,
Rice mode, stereo mode, channel count and sample size the format allows."""
from util import make_file

CASES = [
    dict(name="stereo16-lpc12", channels=2, bps=16, n=4096 * 5 + 1234, bs=4096),
    dict(name="stereo16-fixed", channels=2, bps=16, n=20000, bs=4096, max_lpc_order=0),
    dict(name="mono16", channels=1, bps=16, n=30000, bs=4096),
    dict(name="8ch24-lpc32-po8", channels=8, bps=24, n=16384 * 2 + 100, bs=16384, rate=192000, max_lpc_order=32,
         force_lpc_order=32, force_partition_order=8),
    dict(name="stereo24", channels=2, bps=24, n=20000, bs=4608, rate=96000),
    dict(name="verbatim", channels=2, bps=16, n=5000, bs=1152, force_verbatim=1),
    dict(name="escape-all", channels=2, bps=16, n=6000, bs=1024, escape_mode=2),
    dict(name="escape-some", channels=2, bps=16, n=6000, bs=1024, escape_mode=1),
    dict(name="rice5", channels=2, bps=16, n=6000, bs=1024, rice_method=1),
    dict(name="left-right", channels=2, bps=16, n=6000, bs=1024, stereo_mode=0),
    dict(name="left-side", channels=2, bps=16, n=6000, bs=1024, stereo_mode=1),
    dict(name="side-right", channels=2, bps=16, n=6000, bs=1024, stereo_mode=2),
    dict(name="mid-side", channels=2, bps=16, n=6000, bs=1024, stereo_mode=3),
    dict(name="fixed0", channels=2, bps=16, n=5000, bs=576, force_fixed_order=0),
    dict(name="fixed1", channels=2, bps=16, n=5000, bs=576, force_fixed_order=1),
    dict(name="fixed2", channels=2, bps=16, n=5000, bs=576, force_fixed_order=2),
    dict(name="fixed3", channels=2, bps=16, n=5000, bs=576, force_fixed_order=3),
    dict(name="fixed4", channels=2, bps=16, n=5000, bs=576, force_fixed_order=4),
    dict(name="variable-bs", channels=2, bps=16, n=30000, bs=4096, variable_blocksize=1,
         var_sizes=[1024, 4096, 576, 2000]),
    dict(name="3ch12-bs1000", channels=3, bps=12, n=7777, bs=1000),
    dict(name="mono8", channels=1, bps=8, n=5000, bs=256),
    dict(name="5ch20", channels=5, bps=20, n=9000, bs=2048, rate=48000),
    dict(name="stereo32", channels=2, bps=32, n=5000, bs=512, stereo_mode=0),
    dict(name="lpc32-16bit", channels=2, bps=16, n=9000, bs=4096, max_lpc_order=32),
    dict(name="mono24-odd", channels=1, bps=24, n=5001, bs=1001),
    dict(name="tiny", channels=2, bps=16, n=17, bs=4096),
    dict(name="stereo16-odd-bs", channels=2, bps=16, n=5005, bs=1001),
    dict(name="stereo24-bs1002", channels=2, bps=24, n=5010, bs=1002),
    dict(name="6ch16", channels=6, bps=16, n=7000, bs=1152, rate=48000),
    dict(name="stereo8", channels=2, bps=8, n=5000, bs=512),
    dict(name="mono32", channels=1, bps=32, n=3000, bs=512),
    dict(name="bs65535", channels=2, bps=16, n=65535 + 1234, bs=65535),
    dict(name="bs32768-mono24", channels=1, bps=24, n=32768 + 77, bs=32768),
    dict(name="bs16", channels=2, bps=16, n=1250, bs=16),
    dict(name="bs17-3ch8", channels=3, bps=8, n=1250, bs=17),
]


def encode_case(i, c):
    c = dict(c)
    name = c.pop("name")
    n, bs = c.pop("n"), c.pop("bs")
    f, x = make_file(100 + i, n, c.pop("channels"), c.pop("bps"), c.pop("rate", 44100), bs,
                     c.pop("max_lpc_order", 12), **c)
    return name, f, x
