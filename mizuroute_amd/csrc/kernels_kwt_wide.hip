// The persistent KWT sweep once more, with MZR_KWT_KC_WIDE particle slots per lane of its 4-lane class (kernels_kwt.hip says why it is a
// translation unit of its own): defines mzr_launch_sweep_kwt_wide and nothing else.
#define MZR_KWT_TU_WIDE 1
#include "kernels_kwt.hip"
