#
# MI355X-native re-implementation of the DLWP-CS cubed-sphere hot path.
#
# Same import path and class names as the reference package (DLWP.custom, DLWP.model.DLWPFunctional, DLWP.util), so that
# existing training scripts and saved configs are drop-in; the arithmetic runs in hand-written HIP kernels for gfx950
# (libdlwpcs.so, see include/dlwpcs.h).  See DESIGN.md / INTEGRATION.md at the repository root.
#

__version__ = '0.11.0+mi355x.1'
