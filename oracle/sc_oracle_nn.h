/* sc_oracle_nn.h -- network part of the CPU ORACLE (test infrastructure only; see sc_oracle.h). */
#ifndef SC_ORACLE_NN_H
#define SC_ORACLE_NN_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_net orc_net;

/* number of tensors of ChessModule(n_res_blocks).state_dict() (py/module.py:109-133) */
int orc_net_num_tensors(int n_blocks);
/* shape of tensor t (state_dict order); returns numel */
int64_t orc_net_tensor_shape(int n_blocks, int channels, int t, int shape[4], int* ndim);
/* build-owned deterministic weight generator (same function in tools/scw.py and the engine) */
float orc_prng_weight(uint64_t seed, int tensor, uint64_t idx, double scale, double shift);

/* emulate: 0 fp32 (the parity reference), 1 the engine's bf16 operand rounding, 2 its fp8 (e4m3) mode (nn.c header) */
orc_net* orc_net_create(int n_blocks, int channels, uint64_t seed, int emulate);
/* OCP e4m3 rounding (nearest even, subnormals, clamp to +-448) and the per-output-channel power-of-two weight scale rule */
float orc_e4m3_round(float x);
int orc_fp8_channel_exp(float maxabs);
void orc_net_free(orc_net*);
int orc_net_set_tensor(orc_net*, int t, const float* data, int64_t numel);
int orc_net_get_tensor(const orc_net*, int t, float* out, int64_t numel);

/* ChessModule.forward (eval): boards int8[8][8][112] (rank,file,plane), meta int32[7] ->
 * logp[4672] (log_softmax of the channel-major flattened policy), value (White's view).
 * dbg_latent (optional): trunk output [64][C]. */
void orc_net_forward(const orc_net*, const int8_t* boards, const int32_t* meta, float* logp, float* value,
                     float* dbg_latent);

#ifdef __cplusplus
}
#endif
#endif
