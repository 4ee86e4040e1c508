"""step_amd -- MI355X-native (gfx950) implementation of STeP's DAPS measurement-consistency
hot path behind the reference's own ``forward_operator`` / ``sampler`` interface.

    from step_amd.forward_operator import get_operator      # 'blackhole', 'mri'
    from step_amd.sampler import MCMCSampler, LatentDAPS

Every operator call runs hand-written HIP kernels from ``libstep_amd.so`` (C ABI in
``include/step_amd.h``); there is no CPU or eager-PyTorch fallback.
"""
__version__ = "0.1.0"
