"""ldpc_amd -- MI355X (gfx950) batched LDPC decoding behind the entry points of
mohammedmidhat/LDPC's dec() path.

The product is the C-ABI library ``libldpc_hip.so`` (include/ldpc_hip.h); this
package is the thin ctypes caller the reference's Python prototype
(LDPC_decode.py) would use: ``ldpc_amd.ldpc_decode.run(H, LLR, max_iter)``.
There is no CPU fallback: decoding without the HIP library / a GPU raises.
"""
from .lib import (ALG, CH, DT, SRC, Code, Decoder, GfMonteCarlo, LdpcError, MonteCarlo, lib_path, load_library)  # noqa: F401
from . import alist, ldpc_decode  # noqa: F401

__all__ = ["ALG", "CH", "DT", "SRC", "Code", "Decoder", "GfMonteCarlo", "LdpcError", "MonteCarlo", "alist", "ldpc_decode",
           "lib_path", "load_library"]
