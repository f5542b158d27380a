// Probe (round 5): what does it cost to write 64-byte rows in PIECES from different workgroups?
// A GroupNorm-backward workgroup owns one group = 4 or 8 channels; the planes layout wants [pixel][32 channels] fp16 rows of
// 64 bytes, so a group-per-workgroup kernel can only write 8- or 16-byte pieces of every row, the other groups' workgroups the
// rest.  Variants: piece size 64 (whole rows, baseline) / 16 / 8 bytes; the workgroups that complete a row are either
// `near` = consecutive block ids (round robin over the 8 XCDs: different L2s) or `same` = ids 8 apart (same XCD: one L2 can
// merge the pieces before the line leaves).
//   hipcc -O3 --offload-arch=gfx950 partial_rows.hip -o partial_rows && ./partial_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int PIECE>
__global__ __launch_bounds__(256) void write_pieces(unsigned char* __restrict__ out, long rows, int same_xcd, int rows_per_wg) {
  constexpr int NP = 64 / PIECE;                       // pieces per row = workgroups per row set
  // block id -> (row set, piece)
  long b = blockIdx.x;
  long set; int piece;
  if (same_xcd) {                                      // ids b, b + 8, b + 16, ... share a row set
    const long super = b / (8L * NP), r = b % (8L * NP);
    piece = (int)(r / 8); set = super * 8 + (r % 8);
  } else {
    set = b / NP; piece = (int)(b % NP);
  }
  const long row0 = set * rows_per_wg;
  for (int i = threadIdx.x; i < rows_per_wg; i += 256) {
    const long row = row0 + i;
    if (row >= rows) break;
    unsigned char* p = out + row * 64 + piece * PIECE;
    if (PIECE == 64) {
      uint4 v = make_uint4((unsigned)row, 1, 2, 3);
      reinterpret_cast<uint4*>(p)[0] = v; reinterpret_cast<uint4*>(p)[1] = v;
      reinterpret_cast<uint4*>(p)[2] = v; reinterpret_cast<uint4*>(p)[3] = v;
    } else if (PIECE == 16) {
      *reinterpret_cast<uint4*>(p) = make_uint4((unsigned)row, 1, 2, 3);
    } else {
      *reinterpret_cast<uint2*>(p) = make_uint2((unsigned)row, 1);
    }
  }
}

template <int PIECE>
float run(unsigned char* buf, long rows, int same, int rpw, int reps) {
  constexpr int NP = 64 / PIECE;
  const long sets = (rows + rpw - 1) / rpw;
  const long blocks = ((sets + 7) / 8) * 8 * NP;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(write_pieces<PIECE>, dim3((unsigned)blocks), dim3(256), 0, 0, buf, rows, same, rpw);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(write_pieces<PIECE>, dim3((unsigned)blocks), dim3(256), 0, 0, buf, rows, same, rpw);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  const long bytes = 256L << 20, rows = bytes / 64;
  unsigned char* buf; hipMalloc(&buf, bytes);
  for (int rpw : {1024, 4096}) {
    printf("rows per workgroup %d (%.0f MB written per launch)\n", rpw, bytes / 1e6);
    float t = run<64>(buf, rows, 0, rpw, 20);
    printf("  whole 64-byte rows            : %7.1f us  %5.2f TB/s\n", t * 1e3, bytes / t / 1e9);
    for (int same = 0; same < 2; ++same) {
      t = run<16>(buf, rows, same, rpw, 20);
      printf("  16-byte pieces, %s: %7.1f us  %5.2f TB/s\n", same ? "same XCD     " : "adjacent ids ", t * 1e3, bytes / t / 1e9);
      t = run<8>(buf, rows, same, rpw, 20);
      printf("   8-byte pieces, %s: %7.1f us  %5.2f TB/s\n", same ? "same XCD     " : "adjacent ids ", t * 1e3, bytes / t / 1e9);
    }
  }
  return 0;
}
