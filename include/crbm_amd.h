/*
 * crbm_amd.h -- C-ABI of the MI355X-native CRBM training hot path.
 *
 * This is the drop-in boundary for the path that the reference
 * (schulter/crbm, secomo/convRBM.py) hands to Theano.  The reference has no
 * FFI layer: its inner boundary is the set of compiled-function handles made
 * in CRBM._compileTheanoFunctions (convRBM.py:453-515).  Every entry point
 * below replaces one of those handles (or the shared-variable accessors next
 * to them) one for one; the file:line it replaces is cited per function.
 *
 * Conventions
 *   - plain C, no C++/torch types; all array arguments are HOST pointers to
 *     C-contiguous float32 unless the name ends in _dev;
 *   - tensors use the reference's layouts: visible (n,1,A,L) with A = input_dims (4: DNA), hidden
 *     (n,K,1,Lh), filters (K,1,A,M), bias (1,K), c (1,A); the packed sums of a training step carry
 *     K*A*M weights per block and A letter counts (crbm_sums_count());
 *   - every function returns 0 on success, a negative crbm_status otherwise,
 *     and never throws; crbm_last_error() returns the message;
 *   - calls are synchronous w.r.t. the host unless the name ends in _async;
 *   - one handle owns one GPU (one process per GPU); a handle is not
 *     thread-safe; the library owns all device memory, the caller owns every
 *     host buffer it passes and may free it on return.
 *
 * Visible data must be exactly one-hot (reference sequences.py:28-31 builds
 * it that way); anything else returns CRBM_ERR_NOT_ONEHOT.
 */
#ifndef CRBM_AMD_H
#define CRBM_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRBM_AMD_ABI_VERSION 3

typedef enum crbm_status {
  CRBM_OK = 0,
  CRBM_ERR_INVALID = -1,      /* bad argument / unsupported configuration */
  CRBM_ERR_HIP = -2,          /* a HIP runtime call failed                 */
  CRBM_ERR_NOT_ONEHOT = -3,   /* visible data is not exactly one-hot      */
  CRBM_ERR_NOT_BINARY = -4,   /* hidden state is not exactly 0/1          */
  CRBM_ERR_RCCL = -5,         /* an RCCL call failed / RCCL not loadable   */
  CRBM_ERR_NO_GPU = -6,       /* no usable HIP device                      */
  CRBM_ERR_IPC_TIMEOUT = -7   /* mapped-buffer all-reduce: a peer's sums never arrived (crbm_ipc_*) */
} crbm_status;

/* Hyper-parameters: the reference constructor's arguments
 * (convRBM.py:68-71) plus what Theano kept implicit. */
typedef struct crbm_config {
  int32_t num_motifs;          /* K                                   :111 */
  int32_t motif_length;        /* M  (1..64)                          :112 */
  int32_t input_dims;          /* A: 4 = DNA; 1..64 otherwise (generic kernels) :113 */
  int32_t doublestranded;      /* 0/1                                 :114 */
  int32_t batchsize;           /* number of persistent fantasy chains :115 */
  int32_t cd_k;                /* Gibbs steps per update              :121 */
  int32_t pooling;             /* hidden units compete in groups of this many
                                  positions (1..64; 1 = independent)  :120 */
  int32_t fantasy_hidden_len;  /* hidden length of the fantasy chains;
                                  the reference hard-codes 200        :168 */
  float learning_rate;         /*                                     :116 */
  float momentum;              /*                                     :117 */
  float rho;                   /* resolved target frequency (>0)  :136-140 */
  float lambda_rate;           /*                                     :119 */
  uint64_t seed;               /* Philox key (reference: wall clock   :155) */
  int32_t device;              /* HIP device ordinal                       */
  int32_t reserved;
} crbm_config;

typedef struct crbm_handle crbm_handle;

/* ---- lifetime ------------------------------------------------------------
 * crbm_create replaces CRBM.__init__'s shared-variable setup + the Theano
 * compile step (convRBM.py:127-175): allocates W,b,c, velocities (zero),
 * fantasy chains (zero) and compiles this model's kernels with hiprtc (cached
 * on disk).  W is zero until crbm_set_params. */
int crbm_create(const crbm_config* cfg, crbm_handle** out);
/* Runs only the kernel specialisation step of crbm_create (hiprtc compile of
 * the model's kernels into the on-disk cache); needs no GPU. */
int crbm_precompile(const crbm_config* cfg);
int crbm_destroy(crbm_handle* h);
/* Message of the last failed call on h (h may be NULL: last crbm_create). */
const char* crbm_last_error(const crbm_handle* h);
int crbm_abi_version(void);
/* Number of visible HIP devices (does not initialise a context). */
int crbm_device_count(void);

/* ---- shared-variable accessors (theano.shared get_value/set_value;
 * convRBM.py:133,149,152,158-173, used at :186-188,:233-235) --------------- */
int crbm_set_params(crbm_handle* h, const float* W, const float* b, const float* c);
int crbm_get_params(crbm_handle* h, float* W, float* b, float* c);
int crbm_set_velocities(crbm_handle* h, const float* vW, const float* vb, const float* vc);
int crbm_get_velocities(crbm_handle* h, float* vW, float* vb, float* vc);
/* fantasy_h / fantasy_h_prime, dense (batchsize,K,1,fantasy_hidden_len);
 * h_prime is ignored / may be NULL when single-stranded. */
int crbm_set_fantasy(crbm_handle* h, const float* hid, const float* hid_prime);
int crbm_get_fantasy(crbm_handle* h, float* hid, float* hid_prime);
/* Visible sample of the last Gibbs step, dense (batchsize,1,4,Lf+M-1). */
int crbm_get_fantasy_visible(crbm_handle* h, float* v);
/* Sampler state: key, Gibbs-step counter, evaluation counter, and the global
 * index of this rank's first chain / first data row (data-parallel runs). */
int crbm_set_rng(crbm_handle* h, uint64_t seed, uint32_t gibbs_step, uint32_t eval_step);
int crbm_get_rng(crbm_handle* h, uint64_t* seed, uint32_t* gibbs_step, uint32_t* eval_step);
int crbm_set_shard(crbm_handle* h, uint32_t chain_offset);

/* ---- training ------------------------------------------------------------
 * theano_trainingFct([D]) (convRBM.py:459-464, graph :373-438): one PCD-k
 * SGD+momentum update on mini-batch D (n,1,4,L); mutates W,b,c, velocities,
 * fantasy chains.  n may differ from batchsize (short last slice). */
int crbm_train_step(crbm_handle* h, const float* D, int32_t n, int32_t L);
/* Same, on rows [start,end) of a data set made resident with
 * crbm_dataset_upload (fit() uploads once, convRBM.py:612-615 then only
 * passes slice bounds). */
int crbm_dataset_upload(crbm_handle* h, const float* data, int32_t n, int32_t L);
/* Same from letter codes, one byte per base (0..3 = A,C,G,T: the map of
 * sequences.py:9-17; 0..input_dims-1 for any other alphabet), shape (n,L): 16x less host->device traffic than the
 * float one-hot array that sequences.py:101-117 builds. */
int crbm_dataset_upload_codes(crbm_handle* h, const uint8_t* codes, int32_t n, int32_t L);
/* A handle holds CRBM_DATASET_SLOTS resident data sets (fit() keeps the
 * training set in slot 0 and the test set of convRBM.py:617-625 in slot 1);
 * uploads and every *_resident call address the selected slot (default 0). */
#define CRBM_DATASET_SLOTS 2
int crbm_dataset_select(crbm_handle* h, int32_t slot);
int crbm_train_step_resident(crbm_handle* h, int32_t start, int32_t end);
/* The batch loop of fit() (convRBM.py:612-615) over the selected resident data
 * set: sequential slices of `batchsize` rows (convRBM.py:722-726), one PCD-k
 * update each, enqueued back to back with one host synchronisation at the end.
 * With a communicator, rank r takes rows [n*r/R, n*(r+1)/R) of every slice. */
int crbm_train_epoch_resident(crbm_handle* h, int32_t batchsize);
/* The same loop for a slot that holds only THIS rank's rows of every slice, in
 * slice order (rank r of R owns rows [n*r/R, n*(r+1)/R) of a slice of n rows):
 * every rank uploads 1/R of the `total_rows` data set instead of all of it.
 * `L` is the sequence length of the GLOBAL data set: a rank whose share is empty
 * has no resident rows to read it from, and every rank must normalise the
 * all-reduced sums with the same counts. */
int crbm_train_epoch_sharded(crbm_handle* h, int32_t batchsize, int32_t total_rows, int32_t L);
/* The persistent chain alone (convRBM.py:397-408): k Gibbs steps on all
 * fantasy chains, parameters frozen.  Benchmark entry. */
int crbm_gibbs_steps(crbm_handle* h, int32_t k);
int crbm_gibbs_steps_async(crbm_handle* h, int32_t k);
int crbm_sync(crbm_handle* h);
/* crbm_sync without the (blocking) read-back of the activity monitor: returns when everything launched has completed. */
int crbm_wait_idle(crbm_handle* h);
/* Times `launches` back-to-back launches of k Gibbs steps each with HIP
 * events on the library's stream; returns the total in milliseconds. */
int crbm_time_gibbs(crbm_handle* h, int32_t k, int32_t launches, float* total_ms);
/* Same for full training steps on resident rows [start,end). */
int crbm_time_train(crbm_handle* h, int32_t start, int32_t end, int32_t launches, float* total_ms);

/* ---- stand-alone passes (the graph builders the reference tests compile
 * directly, tests/testcrbm.py:165-171,:226-229) ---------------------------- */
/* _computeHgivenV(data, flip) (convRBM.py:269-275): any of act/prob/sample
 * may be NULL; each is (n,K,1,L-M+1).  Samples use Philox word 3 = rng_step. */
int crbm_h_given_v(crbm_handle* h, const float* v, int32_t n, int32_t L, int32_t flip,
                   uint32_t rng_step, float* act, float* prob, float* sample);
/* _computeVgivenH(h, hprime) (convRBM.py:317-325): hid (n,K,1,Lh) (any
 * finite values), hid_prime NULL or same shape; outputs (n,1,4,Lh+M-1),
 * each may be NULL. */
int crbm_v_given_h(crbm_handle* h, const float* hid, const float* hid_prime, int32_t n,
                   int32_t Lh, uint32_t rng_step, float* act, float* prob, float* sample);

/* ---- evaluation ------------------------------------------------------------
 * theano_getHitProbs (convRBM.py:507-514) -> (n,K,1,L-M+1). */
int crbm_hit_probs(crbm_handle* h, const float* v, int32_t n, int32_t L, float* out);
/* theano_freeEnergy (:501, graph :657-676) -> (n,) */
int crbm_free_energy(crbm_handle* h, const float* v, int32_t n, int32_t L, float* out);
/* theano_fePerMotif (:504, graph :678-697) -> (n,K) */
int crbm_free_energy_per_motif(crbm_handle* h, const float* v, int32_t n, int32_t L, float* out);
/* theano_evaluateData (:487-491) -> mean free energy, mean of a sampled H */
int crbm_eval_data(crbm_handle* h, const float* v, int32_t n, int32_t L, float* mfe, float* nmh);
/* theano_evaluateParams (:494-499) -> rms(W), IC, median IC */
int crbm_eval_params(crbm_handle* h, float* twn, float* ic, float* medic);

/* ---- data-set scale sweeps --------------------------------
 * The same evaluations fed with one byte per base (`_codes`, (n,L), 0..input_dims-1) or
 * with rows [start,end) of the selected resident data set (`_resident`), so
 * that the fp32 one-hot array of sequences.py:101-117 never has to exist.
 * Output pointers of the free-energy calls may be NULL (at least one is set). */
int crbm_hit_probs_codes(crbm_handle* h, const uint8_t* codes, int32_t n, int32_t L, float* out);
int crbm_hit_probs_resident(crbm_handle* h, int32_t start, int32_t end, float* out);
int crbm_free_energy_codes(crbm_handle* h, const uint8_t* codes, int32_t n, int32_t L, float* fe, float* fe_per_motif);
int crbm_free_energy_resident(crbm_handle* h, int32_t start, int32_t end, float* fe, float* fe_per_motif);
int crbm_eval_data_resident(crbm_handle* h, int32_t start, int32_t end, float* mfe, float* nmh);
/* The per-epoch evaluation loop of fit() (convRBM.py:616-625) over the selected resident data set in ONE call:
 * mini-batches of `batchsize` rows exactly as a loop over crbm_eval_data_resident would take them (same sampler
 * steps, one per batch), one host synchronisation.  Returns the mean over batches of the batches' mean free
 * energy and mean sampled hidden activity -- the two numbers of the reference's status line. */
int crbm_eval_epoch_resident(crbm_handle* h, int32_t batchsize, double* mean_fe, double* mean_nmh);
/* The three reductions of theano_getHitProbs' output that the reference's
 * analysis code uses, without materialising (n,K,1,Lh):
 *   hit_max  (n,K)  = P.max(axis=(2,3))    utils.py:154, :242-244
 *   hit_mean (n,K)  = P.mean(axis=(2,3))   utils.py:305
 *   position_mean (K,Lh) = P.mean(axis=(0,2))   utils.py:113-116
 * Any of the three pointers may be NULL. */
int crbm_hit_summary(crbm_handle* h, const float* v, int32_t n, int32_t L, float* hit_max, float* hit_mean,
                     float* position_mean);
int crbm_hit_summary_codes(crbm_handle* h, const uint8_t* codes, int32_t n, int32_t L, float* hit_max,
                           float* hit_mean, float* position_mean);
int crbm_hit_summary_resident(crbm_handle* h, int32_t start, int32_t end, float* hit_max, float* hit_mean,
                              float* position_mean);

/* ---- data-parallel (new: the reference is single-device) -----------------
 * One process per GPU.  Rank 0 calls crbm_comm_unique_id and distributes the
 * 128 bytes by any host channel; every rank then calls crbm_comm_init.  After
 * that crbm_train_step* all-reduces (ncclSum, float32) one packed buffer of
 * raw statistic sums per step over RCCL/xGMI and every rank applies the same
 * update.  Layout of the packed buffer (floats), KAM = K*4*M:
 *   [vh_d KAM][vh_d' KAM][h_d K][h_d' K][sw KAM][sb K][v_d 4][n_d 1]
 *   [vh_m KAM][vh_m' KAM][h_m K][h_m' K][v_m 4][n_m 1]
 * (the primed blocks are present, zero-filled, also when single-stranded). */
#define CRBM_UNIQUE_ID_BYTES 128
int crbm_comm_unique_id(uint8_t id[CRBM_UNIQUE_ID_BYTES]);
int crbm_comm_init(crbm_handle* h, const uint8_t id[CRBM_UNIQUE_ID_BYTES], int32_t nranks, int32_t rank);
int crbm_comm_destroy(crbm_handle* h);
/* ncclBroadcast of W, b, c and the velocities from rank `root`: replicas must
 * start identical because only the statistic sums are reduced afterwards (the
 * reference's randn initialisation, convRBM.py:127-131, differs per process).
 * No-op without a communicator. */
int crbm_comm_broadcast_state(crbm_handle* h, int32_t root);
/* The same all-reduce WITHOUT a collective launch, for the ranks of one node (at most 8): every rank exports a
 * handle of its sums buffer (crbm_ipc_export, hipIpcGetMemHandle), the host ships the handles to all ranks by
 * any channel, every rank maps them (crbm_ipc_attach, `handles` = nranks x CRBM_IPC_HANDLE_BYTES in rank
 * order).  The column reduction of a training step then PUSHES the rank's packed sums into its slot of every
 * rank's buffer and raises its flag there; the update launch of every rank waits for the flags in its own
 * buffer and adds the slots in rank order (bit-identical on all ranks): no launch at all in place of
 * ncclAllReduce, and no read of remote memory on the critical path.  Alternative to crbm_comm_init, not to be
 * combined with it; replicas must start identical (crbm_amd.dist.attach ships rank 0's parameters first).
 * The wait is bounded in time (CRBM_IPC_TIMEOUT_MS, default 30 000, on the GPU's wall clock): once it has run out
 * no further update is applied (the parameters stay those of the last complete step) and every training entry
 * point and crbm_sync return CRBM_ERR_IPC_TIMEOUT; crbm_ipc_status reads the same word (*timed_out != 0).
 * While this form is attached the sums of a step go straight into the mapped buffers: what crbm_train_local-style
 * readers would find in the handle's own sums buffer is not current. */
#define CRBM_IPC_HANDLE_BYTES 64
int crbm_ipc_export(crbm_handle* h, uint8_t handle[CRBM_IPC_HANDLE_BYTES]);
int crbm_ipc_attach(crbm_handle* h, const uint8_t* handles, int32_t nranks, int32_t rank);
int crbm_ipc_detach(crbm_handle* h);
int crbm_ipc_status(crbm_handle* h, int32_t* timed_out);
int crbm_sums_count(const crbm_handle* h);
/* Split form of crbm_train_step for hosts that reduce the sums themselves:
 * local phase -> sums in `sums_out` (host, crbm_sums_count floats);
 * the caller reduces; apply consumes the reduced sums. */
int crbm_train_local(crbm_handle* h, const float* D, int32_t n, int32_t L, float* sums_out);
int crbm_train_apply(crbm_handle* h, const float* sums_in, int32_t L_data);
/* Times `launches` back-to-back all-reduces of the packed sums buffer alone
 * (HIP events on the library's stream; total in milliseconds): the collective's
 * share of a data-parallel training step.  The buffer's contents
 * are scratch at that point of a step; 0 ms without a communicator. */
int crbm_time_allreduce(crbm_handle* h, int32_t launches, float* total_ms);

/* ---- introspection used by bench/profiling ------------------------------- */
typedef struct crbm_launch_info {
  int32_t nq;            /* float4 quads of motifs the kernels are specialised for */
  int32_t group;         /* letters per gather-table group (G)                    */
  int32_t gibbs_grid, gibbs_block, gibbs_seqs_per_tile, gibbs_lds_bytes;
  int32_t stats_grid_x, stats_grid_y, stats_block, stats_lds_bytes;
  int32_t gibbs_sparse;  /* top-down variant of the Gibbs kernel: 1 = walk over set bits (default of every
                            model), 0 = dense tables (CRBM_TOPDOWN=dense, small models); fixed per handle  */
  int32_t activity_ppm;  /* hidden units on per million after the last crbm_sync / crbm_gibbs_steps, -1 = not read yet */
  int32_t stats_fused;   /* 1: the model half of the gradient statistics rides in the Gibbs launch of a training step */
  int32_t chain_parts;   /* plain chain launches (crbm_gibbs_steps*) go out as this many launches of a share of the chains
                            each, on streams of their own (the gibbs_* fields above then describe ONE of them)          */
} crbm_launch_info;
int crbm_get_launch_info(const crbm_handle* h, crbm_launch_info* out);
/* Device-copy bandwidth (float4 copy kernel, HIP events, read + written bytes
 * per second in GB/s) -- the measured ceiling bench.py reports beside the spec. */
int crbm_copy_bandwidth(crbm_handle* h, int64_t bytes, int32_t reps, float* gb_per_s);
/* Shader clock (MHz) during the launches of the last crbm_time_gibbs call, sampled by the chain kernel itself (one
 * block adds its duration in wall-clock ticks and in shader cycles): bench.py prices a step of overlapping launches in
 * shader cycles with it.  0 when unknown. */
int crbm_last_shader_clock(crbm_handle* h, float* mhz);
/* Actual bytes of chain state one Gibbs launch reads+writes in HBM. */
int64_t crbm_gibbs_state_bytes(const crbm_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* CRBM_AMD_H */
