// kvz_inter_tu.hip -- the inter CTU pass's kernel (kvz_inter_kernels.hpp) in a translation unit of its own, compiled by kvazaar_amd/build.py next to kvz_hip.hip.
#include <hip/hip_runtime.h>
#define KVZ_INTER_KERNEL_BODY 1
#include "kvz_inter_kernels.hpp"
