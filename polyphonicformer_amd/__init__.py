"""polyphonicformer_amd -- MI355X-native unified-query decode head for PolyphonicFormer.

Host side (this package): the reference's registry names / constructor kwargs / state_dict keys
(KernelHead, KernelUpdateIterHead, KernelUpdateHead, KernelUpdator) over `libpolyhead.so`
(hand-written gfx950 HIP kernels behind the C ABI in include/polyhead.h)."""
__version__ = "0.1.0"
