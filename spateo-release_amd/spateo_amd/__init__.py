"""spateo_amd: MI355X-native engine for Spateo's SparseVFC morphometric vector-field path (``spateo.tdr``).

    import spateo_amd as st
    st.tdr.morphofield_sparsevfc(adata, ...)      # same signature as spateo.tdr.morphofield_sparsevfc
    st.tdr.morphofield_jacobian(adata)            # ... and the six other morphofield_* evaluators
    st.align.BA_transform(vecfld, points)         # apply a learned alignment field (same kernel)

Only this hot path is implemented; the compute runs in hand-written HIP kernels
(``spateo-release_amd/csrc``) behind the C ABI of ``include/mvf.h``.  There is no CPU fallback.
"""
from . import align, tdr, vectorfield
from ._anndata_lite import AnnDataLite
from .vectorfield import GPVectorField, SparseVFC, SvcVectorField, con_K, set_default_dtype, vector_field_function

__version__ = "0.1.0"
__all__ = ["align", "tdr", "vectorfield", "AnnDataLite", "SparseVFC", "SvcVectorField", "GPVectorField", "con_K", "vector_field_function",
           "set_default_dtype"]
