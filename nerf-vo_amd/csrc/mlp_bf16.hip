// bf16 instantiation of the fully fused MLP kernels (see mlp.hip / mlp_impl.h): every 16-bit stream of the network
// is bf16, the matrix instruction is v_mfma_f32_16x16x16_bf16, accumulation stays fp32 -- BASELINE configs[4]
// ("MFMA bf16 MLP + fp32 hash accumulate").
#define NVO_MLP_BF16 1
#include "mlp_impl.h"
