// The data-gradient chain of the training step's pair MLPs as ONE kernel (pnr_train_chain.hip); launched by
// pnr_render_backward (pnr_train.hip).
#ifndef PNR_TRAIN_CHAIN_H_
#define PNR_TRAIN_CHAIN_H_
#include <stddef.h>
#include <stdint.h>

#include <hip/hip_runtime.h>

namespace pnr {

// weight groups (1 KiB per wave: 4 k-steps of one 32-row output tile) of the chain's four layers, in stream order
constexpr int CNG_B = 8 * 32;   // W3^T: dZ3 [256] -> dG1 [256]
constexpr int CNG_C = 9 * 32;   // W2^T: dZ2 [256] -> [d extras (7 of a 32-row tile) | dH2 [256]]
constexpr int CNG_D = 8 * 32;   // W1^T: dZ1 [256] -> dH1 [256]
constexpr int CNG_E = 8 * 32;   // W0^T: dZ0 [256] -> dX0, the 224 embedding columns in (channel, component) order
constexpr int CNG_TILE = CNG_B + CNG_C + CNG_D + CNG_E;   // 1056
constexpr size_t CHAIN_W_FLOATS = (size_t)CNG_TILE * 256;

struct ChainParams {
    const float *wchain;     // [CNG_TILE groups][64 lanes][4]: k_pack_chain
    const float *w4acc;      // density head in accumulator order: [(tile * 2 + h) * 16 + r]   (k_pack_chain)
    const int *cnt;          // [0] rows = S * K
    const int *row_pidx;     // [rows] neighbour point of the row, -1 = unfilled slot
    const float *row_w;      // [rows] normalised inverse-distance weight
    const float *row_z;      // [rows] density head before the ReLU
    const float4 *d_out;     // [S] .x = d sigma
    const float *XC;         // [S, 256]: d AGG
    const unsigned *tape_bits;        // LeakyReLU masks of H1, H2, G1, G2 as bits (ShadeParams.tape_bits / k_tape_bits)
    size_t bits_rows;
    const float *X0;         // [rows, 288] taped layer-0 inputs (the (sin, cos) pairs of the embedding channels)
    float *D3, *D2, *D1, *D0;         // [rows, 256] gradients at the four pre-activations (A operands of the weight GEMMs)
    float *rowgrad;          // [rows, 40]: d embedding (32) | d colour (3) | d dir (3)
    const int *pt_rank;      // [N + 1] point -> rank among the call's distinct neighbour points
    int *pt_cnt;             // [U] rows per touched point (zeroed by the caller)
    const int *vs_list;      // valid sample -> sample
    const int *smp_ray;      // sample -> ray
    const float *dirs;       // [R, 3]
    float Rw2c[9];
    int K;
};

// dst: CHAIN_W_FLOATS + 256 floats (the chain's weight stream, then w4 in accumulator order)
void launch_pack_chain(const float *w0, const float *w1, const float *w2, const float *w3, const float *w4, float *dst,
                       hipStream_t st);
int launch_pairs_bwd(const ChainParams &P, int64_t rows_max, hipStream_t st);   // PNR_OK or a pnr_status_t
// the mask bits from row-major tapes (a backward that recomputed the MLP chain instead of receiving the render's tape)
void launch_tape_bits(const int *cnt, const float *H1, const float *H2, const float *G1, const float *G2, size_t bits_rows,
                      unsigned *bits, hipStream_t st);

}  // namespace pnr
#endif
