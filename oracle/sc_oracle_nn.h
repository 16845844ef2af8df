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

orc_net* orc_net_create(int n_blocks, int channels, uint64_t seed, int emulate_bf16);
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
