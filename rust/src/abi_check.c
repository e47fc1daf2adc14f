/* compile-time pins of the C ABI the Rust shim relies on (built by build.rs with `cc`) */
#include "sdxl_mi355.h"
_Static_assert(SDXL_OK == 0, "status code");
_Static_assert(SDXL_DTYPE_F32 == 0 && SDXL_DTYPE_F16 == 1 && SDXL_DTYPE_F16_F32RES == 2 && SDXL_DTYPE_F32_SPLIT == 3 && SDXL_DTYPE_F32_SPLIT_MIX == 4 && SDXL_DTYPE_F32_SPLIT_MIX_F16W == 5 && SDXL_DTYPE_F32_SPLIT_MIX_F16W_GEGLU2 == 6 && SDXL_DTYPE_F32_SPLIT_F16W == 7, "dtype codes");
_Static_assert(sizeof(sdxl_conditioning) == 8 * sizeof(void*) + 4 * sizeof(int32_t), "sdxl_conditioning layout");
int sdxl_mi355_abi_check(void) { return SDXL_OK; }
