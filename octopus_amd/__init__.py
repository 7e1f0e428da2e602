"""octopus_amd — MI355X-native pair-HMM haplotype-likelihood engine (Octopus-compatible).

The package holds only what the one hot path needs: csrc/ (HIP kernels + the C-ABI shared library
liboct_phmm.so), abi.py (ctypes mirror of include/oct_phmm.h), engine.py (host-side mirror of the reference's
HaplotypeLikelihoodModel / HaplotypeLikelihoodArray interface over the C ABI) and synth.py (synthetic
workloads of BASELINE.json's configs). Importing the package does not load the GPU library; engine.load()
does, and fails loudly if it is missing.
"""
__version__ = "0.1.0"
