"""waveletsext.jl_amd -- MI355X-native drop-in for the batched wavelet-packet hot path of
WaveletsExt.jl (wpt/iwpt, wpd/iwpd, swpt/swpd, acwpt/acwpd, their `*all` batch drivers and the
JBB best-basis reduction).  Host side: this package mirrors the reference's function names and
argument conventions; compute: hand-written HIP kernels for gfx950 behind the C ABI declared in
include/waveletsext_hip.h (csrc/libwaveletsext_hip.so).  There is no CPU fallback.

The directory name contains a dot, so import it through the `waveletsext_jl_amd` shim at the
repository root (`import waveletsext_jl_amd as wx`).
"""
from .filters import WT, OrthoFilter, ArgumentError, wavelet, daubechies            # noqa: F401
from .util import (maxtransformlevels, isdyadic, ndyadicscales, nodelength, getchildindex,   # noqa: F401
                   getparentindex, getdepth, gettreelength, maketree, isvalidtree, getleaf,
                   getrowrange, getcolrange, main2depthshift, coarsestscalingrange,
                   finestdetailrange, delete_subtree)
from ._arrays import jl_empty, to_device, to_numpy, to_colmajor                     # noqa: F401
from ._lib import WxError, build_info, device_count, set_force_generic, set_host_hugepages, shutdown, LIB_PATH                # noqa: F401
from .dwt import (wpd, wpd_, wpdall, iwpd, iwpd_, iwpdall, wpt, wpt_, iwpt, iwpt_,   # noqa: F401
                  wptall, iwptall, getbasiscoef, getbasiscoefall, dwt, idwt, dwtall, idwtall)
from .swt import (sdwt, sdwt_, sdwtall, isdwt, isdwt_, isdwtall, swpt, swpt_, swptall, iswpt, iswpt_,   # noqa: F401
                  iswptall, swpd, swpd_, swpdall, iswpd, iswpd_, iswpdall)
from .acwt import (acdwt, acdwt_, acdwtall, iacdwt, iacdwt_, iacdwtall, acwpt, acwpt_, acwptall,       # noqa: F401
                   iacwpt, iacwpt_, iacwptall, acwpd, acwpd_, acwpdall, iacwpd, iacwpd_, iacwpdall,
                   autocorr, pfilter, qfilter, make_acqmfpair, make_acreverseqmfpair)
from .bestbasis import (JBB, LoglpCost, NormCost, tree_costs, bestbasistree, bestbasis_treeselection,   # noqa: F401
                        jbb_moments, costs_from_moments, acwpd_jbb_moments,
                        BB, ShannonEntropyCost, LogEnergyEntropyCost, bestbasistreeall)
from .denoising import (HardTH, SoftTH, SemiSoftTH, SteinTH, VisuShrink, SureShrink, RelErrorShrink,   # noqa: F401,E402
                        noisest, threshold, denoise, denoiseall, surethreshold, relerrorthreshold,
                        surethresholdall, relerrorthresholdall)
from .ldb import (TimeFrequency, AsymmetricRelativeEntropy, SymmetricRelativeEntropy, LpDistance,        # noqa: F401,E402
                  HellingerDistance, EarthMoverDistance, ProbabilityDensity, Signatures, SignatureMap, BasisDiscriminantMeasure,
                  FishersClassSeparability, RobustFishersClassSeparability, energy_map,
                  discriminant_measure, discriminant_power, LocalDiscriminantBasis, fit_, fitdec_, transform,
                  fit_transform, inverse_transform, change_nfeatures)
from .siwt import (ShiftInvariantWaveletTransformNode, ShiftInvariantWaveletTransformObject,               # noqa: F401,E402
                   ShiftInvariantWaveletTransformBatch, siwpd, siwpdall, isiwpd, isiwpdall, bestbasistree_,
                   bestbasistreeall_, delete_node_)
from . import siwt as _siwt                                                                                # noqa: E402
_isvalidtree_arrays = isvalidtree                                                                          # noqa: F405


def isvalidtree(*args, **kw):                                                                              # noqa: F811
    """isvalidtree(x, tree) (Wavelets.jl) or isvalidtree(siwtObj[, literal=False]) (siwt/siwt_utls.jl:185-207)"""
    if len(args) == 1 and isinstance(args[0], ShiftInvariantWaveletTransformObject):
        return _siwt.isvalidtree(args[0], **kw)
    return _isvalidtree_arrays(*args, **kw)
