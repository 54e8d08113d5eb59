/* Marlin::prove on N GPUs through the C ABI alone -- no Python, no torch, no launcher.
 *
 * The reference has ONE caller of Marlin::prove (/root/reference src/lib.rs:151-155) and spreads the work over rayon threads
 * (src/ahp/mod.rs:9-10).  Two ways to give that caller N GPUs:
 *
 *   --mode threads   (default) ONE process: mh_group_create makes one context per listed device and joins them with the
 *                    in-process transport (host payloads through shared memory, device buffers pulled peer-to-peer over xGMI);
 *                    mh_group_run runs rank r's work on a thread bound to context r.  What a Rust host does from inside
 *                    `GpuMarlin::prove_sharded` (shim/src/prover.rs).
 *   --mode fork      N processes, one per GPU, over the library's own RCCL communicator: the parent forks N children, rank 0
 *                    draws the ncclUniqueId (mh_rccl_unique_id) and the parent hands its 128 bytes to the others through pipes,
 *                    each child calls mh_init(device) -> mh_marlin_set_rccl(rank, N, id) -> index -> prove and writes its proof
 *                    back through a pipe.  This is what `bench.py --gpus N` needs when torch is absent.
 *
 * Either way every rank indexes and proves the same DummyCircuit instance (benches/bench.rs:26-66 with a = b = 1), and the
 * parent compares the N proofs with the proof of ONE rank working alone: the program exits non-zero on any difference or on
 * any rank's failure.
 *
 *   gcc -std=c99 -O2 -pthread -Iinclude examples/prove_multi_rank.c -Lmarlin_amd -lmarlin_hip -Wl,-rpath,$PWD/marlin_amd -o /tmp/pmr
 *   /tmp/pmr --ranks 8 --devices 0,1,2,3,4,5,6,7 --log 20            # a real node
 *   /tmp/pmr --ranks 4 --devices 0,0,0,0 --log 12                    # four ranks on one GPU (how the tests run it)
 *   MH_RCCL_LIB=tests/mock_rccl/libmock_rccl.so /tmp/pmr --mode fork --ranks 4 --devices 0,0,0,0 --log 12
 *                                                                     # (RCCL itself refuses two ranks on one device)
 */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>
#include "marlin_hip.h"

#define MAX_RANKS 64
#define PROOF_CAP 4096

#define CHECK(call)                                                                                  \
  do {                                                                                               \
    int rc_ = (call);                                                                                \
    if (rc_ != MH_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, mh_last_error()); return rc_; } \
  } while (0)

static const uint64_t FR_ONE[4] = {0x00000001fffffffeull, 0x5884b7fa00034802ull, 0x998c4fefecbc4ff5ull, 0x1824b159acc5056full};
static const uint64_t TAU[4] = {0x243f6a8885a308d3ull, 0x13198a2e03707344ull, 0xa4093822299f31d0ull, 0x082efa98ec4e6c89ull};
static const uint64_t GAMMA[4] = {0x452821e638d01377ull, 0xbe5466cf34e90c6cull, 0xc0ac29b7c97c50ddull, 0x3f84d5b5b5470917ull};

static uint64_t np2(uint64_t n) { uint64_t p = 1; while (p < n) p <<= 1; return p; }

/* universal_setup (known tau) -> index -> prove on the calling thread's current context; the flat proof goes to out */
static int setup_index_prove(unsigned log_n, uint8_t* out, size_t* len_out) {
  const uint64_t nc = 1ull << log_n, ni = 2, rows = nc - 1;
  const uint64_t H = np2(nc), K = np2(3 * nc);
  uint64_t max_degree = 3 * H - 1;
  if (K - 1 > max_degree) max_degree = K - 1;
  uint64_t srs_g, srs_gamma_g, pk;
  CHECK(mh_srs_powers(MH_CURVE_BLS12_381_G1, TAU, FR_ONE, 0, max_degree + 1, &srs_g));
  CHECK(mh_srs_powers(MH_CURVE_BLS12_381_G1, TAU, GAMMA, 0, 3, &srs_gamma_g));
  uint64_t* row_ptr = (uint64_t*)malloc((nc + 1) * sizeof(uint64_t));
  uint32_t* col[3];
  for (uint64_t r = 0; r <= nc; r++) row_ptr[r] = r < rows ? r : rows;
  for (int k = 0; k < 3; k++) {
    col[k] = (uint32_t*)malloc(rows * sizeof(uint32_t));
    for (uint64_t e = 0; e < rows; e++) col[k][e] = k == 0 ? (uint32_t)ni : k == 1 ? (uint32_t)ni + 1 : 1u;
  }
  mh_r1cs_matrices m;
  memset(&m, 0, sizeof(m));
  m.num_constraints = nc; m.num_instance = ni;
  for (int k = 0; k < 3; k++) { m.row_ptr[k] = row_ptr; m.col[k] = col[k]; m.val[k] = NULL; }
  CHECK(mh_marlin_index(&m, srs_g, srs_gamma_g, &pk));            /* every rank indexes in full (the index is never sharded) */
  uint64_t* inst = (uint64_t*)malloc(ni * 32);
  uint64_t* wit = (uint64_t*)malloc((nc - ni) * 32);
  for (uint64_t i = 0; i < ni; i++) memcpy(inst + 4 * i, FR_ONE, 32);
  for (uint64_t i = 0; i < nc - ni; i++) memcpy(wit + 4 * i, FR_ONE, 32);
  uint8_t seed[32];
  for (int i = 0; i < 32; i++) seed[i] = (uint8_t)i;              /* the same zk_rng on every rank: identical arguments everywhere */
  CHECK(mh_marlin_prove(pk, inst, wit, seed, 20, out, PROOF_CAP, len_out));
  CHECK(mh_marlin_pk_free(pk));
  CHECK(mh_bases_free(srs_g));
  CHECK(mh_bases_free(srs_gamma_g));
  free(row_ptr); free(inst); free(wit);
  for (int k = 0; k < 3; k++) free(col[k]);
  return MH_OK;
}

/* ---- mode threads: one process, one context per rank ------------------------------------------------------------------------ */
struct rank_out { unsigned log_n; uint8_t proof[MAX_RANKS][PROOF_CAP]; size_t len[MAX_RANKS]; };
static int rank_main(int rank, void* user) {
  struct rank_out* o = (struct rank_out*)user;
  return setup_index_prove(o->log_n, o->proof[rank], &o->len[rank]);
}
static int run_threads(int n, const int* devs, unsigned log_n, struct rank_out* o) {
  mh_group_t g;
  CHECK(mh_group_create(devs, n, &g));
  o->log_n = log_n;
  int rc = mh_group_run(g, rank_main, o);
  if (rc != MH_OK) fprintf(stderr, "mh_group_run -> %d: %s\n", rc, mh_last_error());
  const int rd = mh_group_destroy(g);
  return rc != MH_OK ? rc : rd;
}

/* ---- mode fork: one process per rank over the library's own RCCL communicator ------------------------------------------------ */
static int read_all(int fd, void* buf, size_t n) { size_t k = 0; while (k < n) { ssize_t r = read(fd, (char*)buf + k, n - k); if (r <= 0) return -1; k += (size_t)r; } return 0; }
static int write_all(int fd, const void* buf, size_t n) { size_t k = 0; while (k < n) { ssize_t r = write(fd, (const char*)buf + k, n - k); if (r <= 0) return -1; k += (size_t)r; } return 0; }
static int child_main(int rank, int n, int dev, unsigned log_n, int fd_id_in, int fd_id_out, int fd_proof) {
  uint8_t id[128];
  CHECK(mh_init(dev));
  if (rank == 0) {                                   /* rank 0 draws the id; the parent relays it */
    CHECK(mh_rccl_unique_id(id));
    if (write_all(fd_id_out, id, 128)) return 90;
  }
  if (read_all(fd_id_in, id, 128)) return 91;
  CHECK(mh_marlin_set_rccl(rank, n, id));            /* collective: returns when this rank's communicator exists */
  uint8_t proof[PROOF_CAP];
  size_t len = 0;
  int rc = setup_index_prove(log_n, proof, &len);
  if (rc != MH_OK) return rc;
  uint64_t l64 = len;
  if (write_all(fd_proof, &l64, 8) || write_all(fd_proof, proof, len)) return 92;
  CHECK(mh_marlin_rccl_destroy());
  CHECK(mh_shutdown());
  return 0;
}
static int run_fork(int n, const int* devs, unsigned log_n, struct rank_out* o) {
  int id_up[2], id_down[MAX_RANKS][2], pr[MAX_RANKS][2];
  pid_t pid[MAX_RANKS];
  if (pipe(id_up)) return 80;
  for (int r = 0; r < n; r++) if (pipe(id_down[r]) || pipe(pr[r])) return 80;
  for (int r = 0; r < n; r++) {
    pid[r] = fork();
    if (pid[r] < 0) return 81;
    if (pid[r] == 0) {                                /* (forked BEFORE any HIP call: the runtime does not survive a fork) */
      int rc = child_main(r, n, devs[r], log_n, id_down[r][0], id_up[1], pr[r][1]);
      _exit(rc == 0 ? 0 : 1);
    }
  }
  uint8_t id[128];
  int bad = read_all(id_up[0], id, 128);
  for (int r = 0; r < n && !bad; r++) bad = write_all(id_down[r][1], id, 128);
  for (int r = 0; r < n; r++) close(pr[r][1]);       /* a child that dies closes its pipe: the read below ends instead of hanging */
  for (int r = 0; r < n; r++) {
    uint64_t l64 = 0;
    if (read_all(pr[r][0], &l64, 8) || l64 > PROOF_CAP || read_all(pr[r][0], o->proof[r], (size_t)l64)) { o->len[r] = 0; bad = 1; }
    else o->len[r] = (size_t)l64;
  }
  for (int r = 0; r < n; r++) {
    int st = 0;
    waitpid(pid[r], &st, 0);
    if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) { fprintf(stderr, "rank %d failed (status %d)\n", r, st); bad = 1; }
  }
  return bad ? 82 : 0;
}

int main(int argc, char** argv) {
  int n = 2, devs[MAX_RANKS], ndev = 0, fork_mode = 0;
  unsigned log_n = 12;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--ranks") && i + 1 < argc) n = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--log") && i + 1 < argc) log_n = (unsigned)atoi(argv[++i]);
    else if (!strcmp(argv[i], "--mode") && i + 1 < argc) fork_mode = !strcmp(argv[++i], "fork");
    else if (!strcmp(argv[i], "--devices") && i + 1 < argc) {
      char* s = argv[++i];
      for (char* t = strtok(s, ","); t && ndev < MAX_RANKS; t = strtok(NULL, ",")) devs[ndev++] = atoi(t);
    } else { fprintf(stderr, "usage: %s [--mode threads|fork] [--ranks N] [--devices d0,d1,...] [--log L]\n", argv[0]); return 64; }
  }
  if (n < 1 || n > MAX_RANKS) { fprintf(stderr, "--ranks must be in [1, %d]\n", MAX_RANKS); return 64; }
  for (int r = ndev; r < n; r++) devs[r] = ndev ? devs[ndev - 1] : 0;
  static struct rank_out multi, solo;
  int rc;
  if (fork_mode) {
    /* the one-rank proof comes from a forked child as well: this process must stay free of HIP before it forks the ranks */
    rc = run_fork(1, devs, log_n, &solo);
    if (rc == 0) rc = run_fork(n, devs, log_n, &multi);
  } else {
    rc = run_threads(1, devs, log_n, &solo);
    if (rc == 0) rc = run_threads(n, devs, log_n, &multi);
  }
  if (rc != 0) { fprintf(stderr, "FAILED (%d)\n", rc); return 1; }
  int same = solo.len[0] > 0;
  for (int r = 0; r < n; r++) same = same && multi.len[r] == solo.len[0] && !memcmp(multi.proof[r], solo.proof[0], solo.len[0]);
  printf("%s: 2^%u constraints, %d ranks on devices", fork_mode ? "fork + native RCCL transport" : "threads + in-process transport", log_n, n);
  for (int r = 0; r < n; r++) printf(" %d", devs[r]);
  printf(": every rank's proof (%zu bytes) %s the one-rank proof\n", solo.len[0], same ? "equals" : "DIFFERS FROM");
  return same ? 0 : 2;
}
