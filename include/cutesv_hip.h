/*
 * cutesv_hip.h — C ABI of libcutesv_hip.so, the MI355X (gfx950) implementation of
 * cuteSV's signature clustering-and-refinement stage and its genotyping re-cluster.
 *
 * Every entry point replaces one piece of the reference's Python interface for this
 * path.  Reference citations are into /root/reference/src/cuteSV/ (v2.1.4):
 *   INDEL = cuteSV_resolveINDEL.py   DUP = cuteSV_resolveDUP.py   INV = cuteSV_resolveINV.py
 *   TRA   = cuteSV_resolveTRA.py     GT  = cuteSV_genotype.py     MAIN = cuteSV (main script)
 *
 * Boundary rules:
 *   - plain pointers and sizes only; the caller owns every host buffer, the library
 *     never retains a host pointer after a call returns;
 *   - device memory belongs to a per-process context (grow-only arena, one HIP stream);
 *   - a context is not thread-safe; distinct contexts are independent; create it
 *     AFTER fork (one HIP device per worker process, as MAIN:1113 uses one task per
 *     pool worker);
 *   - errors are returned as non-zero ints, never abort(); text via csv_last_error().
 *
 * The same structs are consumed by oracle/liboracle.so (csvo_cluster_batch), the CPU
 * restatement used only by tests/, smoke() and bench.py's cpu_baseline leg.
 */
#ifndef CUTESV_HIP_H
#define CUTESV_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSV_ABI_VERSION 8

/* SV types: one (chromosome, type) pair is one segment == one reference pool task
 * (MAIN:1116-1189).  Order of the enum is irrelevant to results. */
enum { CSV_DEL = 0, CSV_INS = 1, CSV_DUP = 2, CSV_INV = 3, CSV_TRA = 4 };

enum {
    CSV_OK = 0,
    CSV_E_INVALID = 1,   /* bad argument / unsupported combination (e.g. TRA + genotype without contig_len) */
    CSV_E_CAPACITY = 2,  /* output arrays too small; n_calls / n_support hold the need */
    CSV_E_HIP = 3,       /* a HIP runtime call failed; see csv_last_error */
    CSV_E_NOMEM = 4,
    CSV_E_UNSORTED = 5,  /* csv_batch_validate: a segment is not in the rebuild order; or CSV_IN_READS_SORTED was promised
                            and a reads block is not sorted by start */
    CSV_E_STATE = 6      /* run/download without a preceding upload */
};

/*
 * One (chromosome, SV type) task.  The scalar fields are the positional arguments of the
 * reference's run_* tuples:
 *   DEL/INS  INDEL:17-18, 222-223  (read_count, threshold_gloab, max_cluster_bias,
 *            minimum_support_reads, action, remain_reads_ratio)
 *   DUP      DUP:17-18   (read_count, max_cluster_bias, sv_size, action, MaxSize)
 *   INV      INV:6-7     (read_count, max_cluster_bias, sv_size, action, MaxSize)
 *   TRA      TRA:30      (read_count, overlap_size, max_cluster_bias, action, gt_round)
 */
typedef struct csv_segment {
    int32_t svtype;             /* CSV_DEL..CSV_TRA */
    int32_t chrom;              /* index into reads_off[] (selects the reads block) */
    int64_t sig_begin;          /* [sig_begin, sig_end) in the signature columns */
    int64_t sig_end;
    int64_t max_cluster_bias;   /* INDEL:61, DUP:35, INV:56, TRA:65 */
    double  diff_ratio;         /* INDEL threshold_gloab (INDEL:138) / TRA overlap_size (TRA:134,211) */
    double  remain_reads_ratio; /* INDEL:46-47,169 (clamped to <= 1 by the library as well) */
    int64_t sv_size;            /* DUP:112 / INV:132 minimum size */
    int64_t max_size;           /* DUP:112 / INV:134 MaxSize, -1 = unlimited */
    int64_t gt_bias;            /* genotype half window: DEL max_cluster_bias (INDEL:103),
                                   INS 1000 (INDEL:312), DUP/INV max_cluster_bias (DUP:72, INV:94),
                                   TRA max_cluster_bias (TRA:164-166) */
    int32_t read_count;         /* min_support */
    int32_t min_support_reads;  /* min(min_support, 5), MAIN:1124 */
    int32_t genotype;           /* the reference's `action` flag */
    int32_t gt_round;           /* TRA only: --gt_round, the iteration cap of count_coverage (GT:62-93) */
} csv_segment;

/*
 * Flat signature / reads columns.  Within a segment the rows are in
 * the reference's sorted, adjacent-deduplicated order (MAIN:764-802, 958-969).
 *   a        int(pos) (DEL/INS)  | pos1 (DUP/INV/TRA)
 *   b        length (DEL/INS)    | pos2 (DUP/INV/TRA)
 *   read_id  interned read name; equal ids <=> equal names
 *   aux      INS: len(inserted sequence) | INV: strand code | TRA: chr2_rank*8 + BND type code (A..D = 0..3,
 *            codes 4..7 = any other BND type:
 *            the library emits nothing for it, TRA:154-155) | DEL/DUP: ignored
 * Reads table (MAIN:733): one block per chromosome (reads_off), rows in ANY order inside a block — the reference's
 * block is the concatenation of per-worker extraction batches, stably sorted by chromosome only (MAIN:810), and
 * overlap_cover sorts its sweep events itself (GT:101-109).  csv_batch_run brings every block into stable start order
 * on the device (stage "reads_order": whole sorted runs are moved when the block is a permutation of disjoint sorted
 * runs, which is what MAIN:697-735 produces; a general stable radix sort otherwise).  CSV_IN_READS_SORTED skips that.
 */
enum {
    CSV_IN_PER_SIG = 1,       /* also produce the per-signature outputs cluster_id / allele_id (csv_batch_download may
                                 then be given those arrays); without it the kernels skip 8 B of stores per signature */
    CSV_IN_READS_SORTED = 2,  /* caller's promise: every reads block is already sorted by r_start (checked on the
                                 device: CSV_E_UNSORTED if not) */
    CSV_IN_SIG_I32 = 4,       /* a and b point to int32_t columns (positions and lengths of a genome fit 31 bits; a third
                                 less data on the link: 67 -> 45 MB for a 30x genome); widened on the device */
    CSV_IN_READS_I32 = 8,     /* r_start and r_end point to int32_t columns */
    CSV_IN_DEVICE_COLUMNS = 16, /* a, b, read_id and aux are DEVICE pointers (memory of this context's GPU, e.g. csv_rebuild_out.dev_* of
                                 a csv_rebuild_signatures call with CSV_RB_KEEP_ON_DEVICE): the columns move device to device
                                 (HBM rate) instead of crossing PCIe twice; the reference's dataflow rebuild -> cluster
                                 (MAIN:750-857 -> 1113-1199) without a host round trip */
    CSV_IN_READS_DELTA16 = 64, /* (ABI v8, with CSV_IN_READS_I32) r_delta / r_len16 (either or both) are given: the reads table's start column
                                 crosses the link as 16-bit gaps (a block is a concatenation of start-sorted runs, MAIN:697-735, 810: a 30x
                                 genome's neighbours are ~0.5 kb apart) and its end column as 16-bit lengths (HiFi reads are < 64 kb), both
                                 rebuilt on the device: 13 -> 9 bytes per read, the bulk of a genotyping call's upload */
    CSV_IN_SIG_DELTA16 = 32   /* (ABI v8, with CSV_IN_SIG_I32, host columns) a_delta / a_esc_* are given: the position column crosses
                                 the link as 16-bit gaps - the rebuild order (MAIN:764-802) makes it non-decreasing inside a segment,
                                 a 30x genome's neighbours are ~1 kb apart - and is rebuilt on the device (k_unpack_a16): 11.1 -> 5.6 MB
                                 for a 30x genome, the largest single transfer of a gate-first call.  `a` must still be given (the
                                 library reads a few rows of it on the host; every other path uses it as before) */
};
typedef struct csv_batch_in {
    int32_t            n_seg;
    int32_t            n_chrom;
    const csv_segment* seg;
    int64_t            n_sig;
    const int64_t*     a;
    const int64_t*     b;
    const int32_t*     read_id;
    const int32_t*     aux;
    const int64_t*     reads_off;   /* n_chrom + 1 offsets; NULL when no segment genotypes */
    int64_t            n_reads;
    const int64_t*     r_start;
    const int64_t*     r_end;
    const uint8_t*     r_primary;
    const int32_t*     r_id;
    const int64_t*     contig_len;  /* n_chrom reference lengths (bamfile.get_reference_length, TRA:264,291); NULL unless a
                                       TRA segment genotypes */
    int32_t            flags;       /* CSV_IN_* */
    int32_t            reserved;
    /* (ABI v8) CSV_IN_SIG_DELTA16: a_delta[i] = a[i] - a[i - 1] where that lies in [0, 0xFFFF) and i > 0, else 0xFFFF and row i is listed
     * in a_esc_row (ascending) with a_esc_val = a[i].  n_sig entries; built once per store (SigStore.pinned()).  Rows that begin a
     * segment or a chain tile need no escape: the library anchors them itself.  NULL / 0 without the flag. */
    const uint16_t*    a_delta;
    int64_t            n_esc;
    const int64_t*     a_esc_row;
    const int32_t*     a_esc_val;
    /* (ABI v8, optional, with CSV_IN_SIG_I32) the length and read-id columns once more, interleaved: rows8[2 * i] = b[i],
     * rows8[2 * i + 1] = read_id[i], in page-locked memory.  A gate-first csv_cluster_batch (csv_batch_info 0) then reads the rows of
     * the clusters that pass the size gate (INDEL:62-64, 86) out of THIS array: a run of ~20 rows is 3-4 PCIe lines of 64 bytes
     * instead of 2-3 in each of two columns (the fetch is the largest item of such a call).  b and read_id must still be given. */
    const int32_t*     rows8;
    /* (ABI v8) CSV_IN_READS_DELTA16.  r_delta[i] = r_start[i] - r_start[i - 1] where that lies in [0, 0xFFFF) and i > 0, else 0xFFFF and
     * row i is listed in r_esc_row (ascending) with r_esc_val = r_start[i] (the first row of every sorted run is one).  r_len16[i] =
     * r_end[i] - r_start[i] where that lies in [0, 0xFFFF), else 0xFFFF and row i is listed in l_esc_row (ascending) with l_esc_val =
     * r_end[i].  n_reads entries each; r_start / r_end must still be given (the library reads a few rows of r_start on the host).
     * Either pointer may be NULL.  A column that is mostly escapes (shuffled blocks; ultra-long reads for r_len16) travels as itself. */
    const uint16_t*    r_delta;
    int64_t            n_r_esc;
    const int64_t*     r_esc_row;
    const int32_t*     r_esc_val;
    const uint16_t*    r_len16;
    int64_t            n_l_esc;
    const int64_t*     l_esc_row;
    const int32_t*     l_esc_val;
    /* (ABI v8, optional) r_idp[i] = r_id[i] | r_primary[i] << 31: the read id and the primary flag in one word (the form the device
     * keeps them in anyway).  When given, r_primary / r_id do not cross the link (5 -> 4 bytes per read); both must still be valid
     * pointers for the paths that read them on the host. */
    const uint32_t*    r_idp;
} csv_batch_in;

/*
 * Caller-allocated structure-of-arrays result.  One entry per candidate SV ("call"), in the
 * reference's emission order: segments in input order, clusters in file order, alleles /
 * sub-clusters in the order the reference appends them (INDEL:163-219, DUP:95-131, INV:124-203,
 * TRA:131-254).  Field meaning per type:
 *             bp1                  bp2                    support             search_pos
 *   DEL   int(breakpointStart)  int(signalLen) (>0)   allele read count    search_threshold (INDEL:177)
 *   INS   int(breakpointStart)  int(signalLen)        allele read count    = bp1 (INDEL:415)
 *   DUP   breakpoint_1          breakpoint_2          unique reads         -
 *   INV   breakpoint_1          breakpoint_2          unique reads         -
 *   TRA   int(sum p1 / n)       int(sum p2 / n)       unique reads         -
 * cipos / cilen: the integer inside cal_CIPOS's "-%d,%d" (GT:58-60), DEL/INS only.
 * seq_pick: INS only, global signature index whose sequence is sliced for ALT (INDEL:399-403).
 * call_aux: aux of the cluster's first signature (INV strand / TRA chr2,type).
 * dr / dv / gl_idx: genotype read counts (GT:161-173) and the index of cal_GL's result
 *   (GT:33-56) in the table defined by csv_gl_index(); -1 when the segment is not genotyped.
 *   TRA: call_gt / count_coverage (TRA:258-309, GT:62-93) evaluated over the READS TABLE
 *   instead of a BAM re-fetch: `fetch(chr, s, e)` = the reads of the chromosome's block with start < e and
 *   end > s, in block order; flag in [0, 16] = r_primary.  Identical to the reference whenever the BAM holds
 *   no alignment inside the two windows that the reads table drops (secondary flag 256/272, mapq < min_mapq,
 *   MAIN:711-733).  count_coverage's give-up status (-1) is dr = -1, gl_idx = -1 ("./.", DR ".").
 * support_off/support_sig: CSR list of the signatures whose read names form the call's
 *   read list, in reference order where that order is deterministic.
 * cluster_id / allele_id: optional per-signature outputs (NULL to skip; they need CSV_IN_PER_SIG on the batch,
 *   csv_cluster_batch sets it by itself when either array is given): dense id of the chained cluster a signature
 *   belongs to, and the index of the call (0-based, global) it supports or -1.
 * seg_status: optional (NULL to skip) n_seg words of CSV_SEG_* bits.  The reference has no size limits
 *   (INDEL:110-136, GT:95-159) and neither has this library; the only per-segment condition left is a length / pos2
 *   value outside [0, 2^42) (the clusters holding one emit nothing, every other cluster of the batch is unaffected;
 *   main_ctrl swallows a failing task the same way, MAIN:1193-1199).
 * (ABI v7) A caller that does not need a field does not pay for its trip across PCIe: call_cluster, call_aux, cipos, cilen,
 *   search_pos, seq_pick, dr, dv and gl_idx may each be NULL (not written); call_seg, bp1, bp2 and support are always required.
 *   `flags`: CSV_OUT_NO_SUPPORT_LIST - support_off / support_sig / support_sig32 are not written and may be NULL (`support` still
 *   holds every call's read count, n_support the total): read names reach the VCF only under --report_readid (GT:263-458), and
 *   the list is half of a discovery run's result bytes.  CSV_OUT_COORD_I32 - bp1, bp2, search_pos and seq_pick point to int32_t
 *   arrays (only with CSV_IN_SIG_I32 columns, whose coordinates fit by construction; CSV_E_INVALID otherwise).
 */
enum { CSV_SEG_KEY_RANGE = 1 };
enum { CSV_OUT_NO_SUPPORT_LIST = 1, CSV_OUT_COORD_I32 = 2 };
typedef struct csv_batch_out {
    int64_t  cap_calls;
    int64_t  cap_support;
    int64_t  n_calls;       /* out */
    int64_t  n_support;     /* out */
    int64_t  n_clusters;    /* out: chained clusters over the whole batch */
    int32_t* call_seg;
    int32_t* call_cluster;
    int32_t* call_aux;
    int64_t* bp1;
    int64_t* bp2;
    int32_t* support;
    int32_t* cipos;
    int32_t* cilen;
    int64_t* search_pos;
    int64_t* seq_pick;
    int32_t* dr;
    int32_t* dv;
    int32_t* gl_idx;
    int64_t* support_off;   /* cap_calls + 1 */
    int64_t* support_sig;   /* cap_support */
    int32_t* cluster_id;    /* n_sig or NULL */
    int32_t* allele_id;     /* n_sig or NULL */
    int32_t* seg_status;    /* n_seg or NULL */
    int32_t* support_sig32; /* (ABI v6) cap_support or NULL: the support list as int32 - a batch holds fewer than 2^31 signatures -
                               INSTEAD of support_sig (exactly one of the two is given): half the bytes of the largest result
                               array on the link */
    int32_t  flags;         /* (ABI v7) CSV_OUT_* */
    int32_t  reserved;
} csv_batch_out;

/* Per-kernel device timings of one csv_batch_run, measured with HIP events recorded on the
 * context's stream around every launch (only when stats != NULL; the plain run records nothing).
 * Kernel names: csv_stage_name(i); unused slots are 0. */
#define CSV_N_STAGES 24
typedef struct csv_run_stats {
    float   ms_total;
    float   ms_stage[CSV_N_STAGES];
    int64_t n_clusters;
    int64_t n_work_wave;    /* clusters refined by the one-wavefront-per-cluster tier */
    int64_t n_work_block;   /* clusters refined by the workgroup (LDS / global scratch) tier */
    int64_t n_calls;
    int64_t n_support;
} csv_run_stats;

typedef struct csv_ctx csv_ctx;

int         csv_abi_version(void);
/* sizeof() of the ABI's structs as this library was compiled (which: 0 csv_segment, 1 csv_batch_in, 2 csv_batch_out,
 * 3 csv_run_stats, 4 csv_rebuild_in, 5 csv_rebuild_out, 6 csv_vcf_in, 7 csv_rows_in, 8 csv_cigar_in, 9 csv_cigar_out, 10 csv_split_in,
 * 11 csv_split_out; -1 otherwise): lets a binding
 * check its own mirror of the layouts at load time. */
int         csv_struct_size(int which);
int         csv_device_count(int* n);
/* (ABI v7) PCI bus id of a device ("0000:c1:00.0") and its compute-unit count: what a multi-process launcher logs per rank to show
 * that N ranks sit on N distinct GPUs (the reference's pool workers have no placement: MAIN:1113).  Returns CSV_OK. */
int         csv_device_info(int device_id, char* pci_bus_id, int cap, int* n_cu);
int         csv_ctx_create(int device_id, csv_ctx** out);
void        csv_ctx_destroy(csv_ctx* ctx);
const char* csv_last_error(const csv_ctx* ctx);
const char* csv_stage_name(int stage);

/* One-shot: H2D, all kernels, D2H.  Replaces the bodies of resolution_DEL/INS/DUP/INV/TRA
 * (INDEL:17-108, 222-317; DUP:17-77; INV:6-99; TRA:30-104) and of call_gt -> overlap_cover
 * -> assign_gt (INDEL:441-479, DUP:137-181, INV:208-252, GT:95-173) for every segment of
 * the batch at once. */
int csv_cluster_batch(csv_ctx* ctx, const csv_batch_in* in, csv_batch_out* out);

/* Resident mode: the same work split at the PCIe boundary, so a caller (or bench.py) can
 * keep the columns in HBM and run the kernels repeatedly. */
int csv_batch_upload(csv_ctx* ctx, const csv_batch_in* in);
/* csv_batch_run only enqueues work (it returns before the kernels finish; csv_batch_download / csv_ctx_sync wait) and never
 * waits for the device.  A run of an upload that has been run before launches the refine tiers above 64 signatures only if
 * that earlier run reported clusters of that size (one page-locked word the device writes; same columns and parameters
 * give the same tiers); CSV_NO_PEEK=1 in the environment launches every tier always. */
int csv_batch_run(csv_ctx* ctx, csv_run_stats* stats /* nullable */);
int csv_batch_download(csv_ctx* ctx, csv_batch_out* out);
int csv_ctx_sync(csv_ctx* ctx);
/* (ABI v7) Pipelined delivery for a resident caller that runs batch after batch (or one batch again and again): the result of run
 * k is written into the caller's page-locked arrays by the device (as csv_batch_download does) on a stream of its own WHILE run
 * k + 1 computes - consecutive runs alternate between two result arenas on the device.  csv_batch_publish_async starts the
 * delivery of the LAST run's result into `out` and returns at once; csv_batch_publish_wait blocks until the OLDEST delivery in
 * flight is complete, fills n_calls / n_support / n_clusters / seg_status of its result struct - returned through `done` - and
 * returns its status (CSV_E_CAPACITY etc.).  At most two deliveries in flight, each into arrays of its own; every array must be
 * page-locked; no per-signature outputs; the upload must have been downloaded once synchronously before (that settles how its
 * reads table is ordered).  The sequence  run, async(A), run, async(B), wait -> A, run, async(A), wait -> B ...  keeps the link
 * busy under the kernels.  HOW the result crosses depends on where the caller put the arrays: arrays that sit exactly back to back
 * in page-locked memory (at most three such runs: e.g. every per-call array and the support list carved out of ONE csv_host_alloc
 * block, no gaps) are written as a device image behind the run's kernels and moved by the copy engine, one transfer per run of
 * arrays - in full, i.e. cap_calls / cap_support elements each: size them from the first download; scattered arrays are written in
 * place by a kernel on a stream of its own.  The first form is the one that overlaps: a kernel storing across PCIe holds up every
 * kernel boundary of the run beside it.  Elements beyond n_calls / n_support are unspecified in both forms.
 * No counterpart in the reference (its workers return pickled rows through a pipe, MAIN:1191-1197). */
int csv_batch_publish_async(csv_ctx* ctx, csv_batch_out* out);
int csv_batch_publish_wait(csv_ctx* ctx, csv_batch_out** done /* nullable */);
/* How the reads table of the last completed run was brought into start order: 0 = the caller promised sorted blocks,
 * 1 = whole sorted runs were moved (or nothing had to move), 2 = the general stable radix sort; -1 = no reads table. */
int csv_batch_reads_mode(const csv_ctx* ctx);
/* (ABI v7) What the last upload of this context did at the PCIe boundary: which = 0: 1 when csv_cluster_batch took the
 * gate-first form (page-locked b / read_id / aux columns: only the position column travels in bulk, the rows of the clusters
 * that pass the size gate - INDEL:62-64, 86 - are read out of the caller's columns by the device), else 0; which = 1: the
 * bytes of signature columns the bulk copy therefore did not send (ABI v8: including the half of the position column that
 * CSV_IN_SIG_DELTA16 saves); which = 2 (ABI v8): 1 when the position column crossed as 16-bit gaps; which = 3 (ABI v8): bit 0 / bit 1 when the reads table's
 * starts / ends crossed as 16-bit gaps / lengths.  A measurement aid
 * (bench.py's pcie object). */
int csv_batch_info(const csv_ctx* ctx, int which, int64_t* value);
/* Context options.  CSV_OPT_REUSE_READS_ORDER (default 1): the start-ordered, packed copy of the reads table that the first
 * csv_batch_run after an upload builds is kept for later runs of the SAME upload (a resident caller that re-runs a batch,
 * e.g. with other segment scalars, does not re-order millions of reads every time); 0 rebuilds it in every run.
 * csv_cluster_batch always builds it (every call is a new upload). */
enum { CSV_OPT_REUSE_READS_ORDER = 1 };
int csv_batch_option(csv_ctx* ctx, int option, int value);

/* Page-locked host memory.  Columns that live in it (or in a registered caller buffer) are copied asynchronously, so the
 * kernels start while the later columns are still on the link; a csv_cluster_batch call whose b / read_id / aux columns live in
 * it copies only the position column and lets the device read the rows it needs from the others (gate-first; batches of at
 * least CSV_LAZY_MIN = 65 536 signatures, CSV_NO_LAZY=1 in the environment switches it off); RESULT arrays that live in it are filled in place by the
 * device (csv_batch_download / csv_cluster_batch then cost one synchronisation: no staging copy, no host-side unpack) -
 * all of bp1 ... support_sig must be page-locked for that, seg_status and the per-signature arrays may be anywhere.
 * Memory from csv_host_alloc / csv_host_register must be released through csv_host_free / csv_host_unregister.  No
 * counterpart in the reference (its pool workers unpickle tuples, INDEL:52-58).  csv_host_alloc needs no context; the
 * memory is usable from every context of the process. */
int  csv_host_alloc(int64_t bytes, void** out);
void csv_host_free(void* p);
int  csv_host_register(void* p, int64_t bytes);
int  csv_host_unregister(void* p);

/* Optional check of the input order contract on the uploaded batch: inside every segment the rows must be
 * strictly increasing in the reference's rebuild sort key (cuteSV main script :764-802; adjacent duplicates
 * removed, :958-969).  Returns CSV_E_UNSORTED otherwise.  One pass over the columns; not part of csv_batch_run. */
int csv_batch_validate(csv_ctx* ctx);

/* Measurement aid (bench.py's roofline object): device-to-device copy bandwidth of this GPU, read + write bytes
 * over the best of `reps` hipMemcpyAsync calls of `bytes` bytes, in GB/s.  No counterpart in the reference. */
int csv_measure_copy_bandwidth(csv_ctx* ctx, int64_t bytes, int reps, double* gb_per_s);
/* Measurement aid: evict the GPU's caches (per-XCD L2s and the 256 MiB Infinity Cache) by overwriting a scratch buffer of
 * `bytes` bytes on the context's stream (allocated on first use), so that the next csv_batch_run finds its columns in
 * HBM only: the "cold" figures of bench.py.  No counterpart in the reference. */
int csv_cache_flush(csv_ctx* ctx, int64_t bytes);

/* cal_GL's domain after its special cases and rescale_read_counts (GT:25-37): returns the
 * table index the device writes into gl_idx for (DR, DV) = (c0, c1).  Host-side helper so
 * the Python shim and the tests share one definition with the kernels. */
int32_t csv_gl_index(int64_t c0, int64_t c1);
#define CSV_GL_TABLE_SIZE (101 * 101 + 2)

/* ---------------------------------------------------------------------------------------------
 * Rebuild step on the GPU: unsorted signature rows -> the order contract of
 * process_process_sigs_type (cuteSV main script :750-857): rows sorted by
 *   (segment, [aux for INV / TRA segments], a, b, read_id)           (:764-802 sort keys)
 * with adjacent exact duplicates removed (:958-969).  `seg_id` is the caller's ordinal of the row's
 * (SV type, chromosome) pair in the order the segments should come out; seg_aux_major[s] = 1 for INV and
 * TRA segments (strand / chr2,type sort before the position).  Stable LSD radix sort of a row permutation,
 * 8 bits per pass, zero bytes skipped; then gather + de-duplication.  Outputs are caller-allocated with
 * room for n rows; src_row[i] = input row of output row i (to carry payloads such as INS sequences).
 */
enum {
    CSV_RB_KEEP_ON_DEVICE = 1,       /* csv_rebuild_in.flags: the sorted columns stay in device memory (csv_rebuild_out.dev_*, valid
                                        until the context's next csv_rebuild_signatures / csv_cigar_signatures / csv_split_signatures
                                        call); host output arrays that are NULL are not written */
    CSV_RB_FROM_POOL = 2             /* the rows are the context's device-resident signature pool (below): n, seg_id, a, b, read_id and
                                        aux of csv_rebuild_in are ignored; a row's read index is replaced by read_rank[index] (the
                                        rank of the read's NAME: string order is the caller's business); src_row numbers pool rows */
};
/* The rows of the keep-every-row segments (seg_nodedup) whose integer keys tie - (segment, a, b, read_id) equal - are ordered
 * by data only the caller has: the reference sorts INS rows by (chr, int(pos), len, read, SEQUENCE) and drops a row only when
 * the whole tuple repeats, the x.5 of a split-read position included (MAIN:774-775, :958-969).  csv_rebuild_signatures sorts on
 * the integer columns on the device, hands the tie groups - a few rows per genome - to this function ONCE, on the calling
 * thread, and applies the answer to the device-resident permutation before the rows are gathered: the columns never come to
 * the host for it (CSV_RB_KEEP_ON_DEVICE -> CSV_IN_DEVICE_COLUMNS stays intact).
 *   rows [group_off[g], group_off[g + 1]) of src_row are group g, in the device's stable order (= input order);
 *   order[i]  the callee's position of row i INSIDE its group (a permutation of 0 .. size - 1 per group),
 *   drop[i]   1: the row is a duplicate and is removed.
 * Return 0; anything else fails the call with CSV_E_INVALID. */
typedef int (*csv_tie_order_fn)(void* user, int64_t n_groups, const int64_t* group_off, const int32_t* src_row, int32_t* order, uint8_t* drop);

typedef struct csv_rebuild_in {
    int64_t        n;
    int32_t        n_seg;
    int32_t        flags;           /* CSV_RB_* */
    const uint8_t* seg_aux_major;   /* n_seg */
    const int32_t* seg_id;
    const int64_t* a;
    const int64_t* b;
    const int32_t* read_id;
    const int32_t* aux;
    const uint8_t* seg_nodedup;     /* n_seg or NULL: 1 = sort this segment but keep every row.  INS rows are equal only when
                                       their sequences and the x.5 of a split-read position are equal too (MAIN:228, :774-775):
                                       tie_order (below) settles those few groups; without it the caller finishes them on the host */
    const int32_t* read_rank;       /* CSV_RB_FROM_POOL: n_rank ranks, indexed by the pool rows' read index */
    int64_t        n_rank;
    csv_tie_order_fn tie_order;     /* nullable (ABI v6): see csv_tie_order_fn */
    void*          tie_user;
} csv_rebuild_in;

typedef struct csv_rebuild_out {
    int64_t  n_out;                 /* out */
    int32_t* seg_id;
    int64_t* a;
    int64_t* b;
    int32_t* read_id;
    int32_t* aux;
    int32_t* src_row;
    float    ms_device;             /* out: kernels only (HIP events) */
    int32_t  n_passes;              /* out: radix passes executed */
    int64_t* seg_count;             /* n_seg or NULL: rows of every segment after the de-duplication (the csv_segment ranges of the
                                       sorted columns follow from these by a prefix sum) */
    int64_t  n_ins_ties;            /* out: rows of seg_nodedup segments that agree with their predecessor in (segment, a, b, read_id) and
                                       were NOT settled by tie_order: 0 means the device order is final, otherwise (no tie_order given)
                                       the caller finishes those groups on the host */
    void*    dev_seg_id;            /* out (CSV_RB_KEEP_ON_DEVICE): device addresses of the sorted columns, n_out rows each: */
    void*    dev_a;                 /*   int32 seg_id, int64 a, int64 b, int32 read_id, int32 aux, int32 src_row */
    void*    dev_b;
    void*    dev_read_id;
    void*    dev_aux;
    void*    dev_src_row;
    int64_t  n_tie_rows;            /* out (ABI v6): rows handed to tie_order, and how many of them it dropped */
    int64_t  n_tie_dropped;
} csv_rebuild_out;

int csv_rebuild_signatures(csv_ctx* ctx, const csv_rebuild_in* in, csv_rebuild_out* out);

/* The device-resident signature pool: the reference's dataflow extraction -> rebuild (MAIN:697-743 -> 750-857) without a trip
 * through host memory.  csv_cigar_signatures with CSV_CG_TO_POOL appends the signatures it finds as rows (segment, position,
 * length, global read index, aux), csv_split_signatures its candidates; csv_pool_append adds rows the host made;
 * csv_rebuild_signatures with CSV_RB_FROM_POOL sorts and de-duplicates the pool.  The pool belongs to the context and
 * lives until csv_pool_reset / csv_ctx_destroy. */
int csv_pool_reset(csv_ctx* ctx);
int csv_pool_rows(const csv_ctx* ctx, int64_t* n_rows);
int csv_pool_append(csv_ctx* ctx, int64_t n, const int32_t* seg_id, const int64_t* a, const int64_t* b, const int32_t* read, const int32_t* aux);

/* ---------------------------------------------------------------------------------------------
 * The CIGAR scan of the extraction step on the GPU.  Restates the CIGAR part of parse_read
 * (cuteSV main script :606-655: every I / D operation of at least min_siglength bases is a piece at the reference position
 * it is reached, :629-643) and generate_combine_sigs (:515-575: pieces of one type within merge_ins_threshold /
 * merge_del_threshold of each other inside a read become one signature; the distance rule of :535 / :558 / :569 is
 * reproduced as written).  BAM decode and the bases stay in the Python driver with pysam (the SA-tag split-read analysis
 * is csv_split_signatures below): the input is the flat BAM-encoded CIGAR array of a batch of reads (pysam: read.cigartuples), the output the
 * INS / DEL signatures in read order plus, for INS, the query slices the inserted sequence is made of
 * (query_sequence[qoff : qoff + len] per piece, concatenated: :639-640, :537).
 *   use[r] = 0 skips read r (mapq < min_mapq, :614; the query_length < min_read_len gate of :607 is the caller's too).
 * Outputs are caller-allocated; CSV_E_CAPACITY fills n_sig_ins / n_piece_ins / n_sig_del with the need.
 */
typedef struct csv_cigar_in {
    int64_t         n_reads;
    const int64_t*  cig_off;        /* n_reads + 1 offsets into cigar */
    const uint32_t* cigar;          /* oplen << 4 | op, op = 0..9 for M I D N S H P = X B (the BAM encoding) */
    const int64_t*  ref_start;      /* read.reference_start, 0-based */
    const uint8_t*  use;            /* n_reads or NULL */
    int32_t         min_siglength;  /* --min_siglength (cuteSV_Description.py:152) */
    int32_t         flags;          /* CSV_CG_* */
    int64_t         merge_ins_threshold;   /* --merge_ins_threshold (:127) */
    int64_t         merge_del_threshold;   /* --merge_del_threshold (:123) */
    /* CSV_CG_TO_POOL: the signatures also become rows of the context's pool - INS rows in segment seg_ins, DEL rows in seg_del
     * (the caller's numbering of (chromosome, type)), read index = read_base + the read's index in this batch, aux of an INS
     * row = length of the sequence the caller would cut out of the read (the pieces clipped to query_len[read] like Python
     * slices; query_len NULL: the signature's length).  Output arrays of csv_cigar_out that are NULL are then not written
     * (their capacities still bound the counts). */
    int32_t         seg_ins, seg_del;
    int64_t         read_base;
    const int32_t*  query_len;      /* n_reads or NULL */
} csv_cigar_in;
enum { CSV_CG_TO_POOL = 1 };

typedef struct csv_cigar_out {
    int64_t  cap_sig_ins, cap_piece_ins, cap_sig_del;
    int64_t  n_sig_ins, n_piece_ins, n_sig_del;          /* out */
    int32_t* ins_read;  int64_t* ins_pos;  int64_t* ins_len;  int64_t* ins_piece0;  int32_t* ins_npiece;     /* cap_sig_ins */
    int32_t* piece_qoff;  int32_t* piece_len;                                                               /* cap_piece_ins */
    int32_t* del_read;  int64_t* del_pos;  int64_t* del_len;                                                /* cap_sig_del */
    float    ms_device;                                  /* out: kernels only (HIP events) */
    int32_t  reserved;
} csv_cigar_out;

int csv_cigar_signatures(csv_ctx* ctx, const csv_cigar_in* in, csv_cigar_out* out);

/* ---------------------------------------------------------------------------------------------
 * The split-read analysis of the extraction step on the GPU (second half).  Restates
 * organize_split_signal (cuteSV main script :483-513: the primary alignment plus the SA-tag entries that pass min_mapq
 * become [read_start, read_end, ref_start, ref_end, chr, strand] segments; reads with more than max_split_parts segments
 * are skipped) and analysis_split_read with analysis_inv / analysis_bnd (:50-464: segments sorted by read_start, then
 * the two-segment and sliding three-segment rules that emit INV / TRA / DUP / INS / DEL candidates).  BAM decode and
 * the text of the SA tag stay in the Python driver with pysam (north_star): per entry the caller passes what
 * acquire_clip_pos (:466-481) takes out of the SA entry's CIGAR and the numbers of the entry itself.
 *
 * Entries of read r: [ent_off[r], ent_off[r + 1]), the primary alignment first when the read has one (parse_read
 * :660-668, primary = 1: c0/c1 = read_start/read_end, f0/f1 = ref_start/ref_end as parse_read computes them), then the
 * SA entries in tag order (primary = 0: c0/c1 = leading / trailing soft-clip lengths, f0 = 0-based start, f1 = reference
 * span, mapq).  chr is an integer whose order is the Python string order of the chromosome names (analysis_bnd compares
 * names with <, :110).  strand: 0 '+', 1 '-'.
 *
 * Output: one record per candidate, reads in order, inside a read in the order the reference appends them (per SV type
 * that is exactly the order of the reference's five lists).  kind: 0 DEL (a = pos, b = length), 1 INS (a = position
 * numerator, b = length, c / d = the Python slice bounds query[c:d] of the inserted sequence, aux bit 0: the slice is
 * taken from the reverse complement of the query analysis_split_read was given, aux bit 1: the position is the float
 * a / 2 (:228, :244) rather than the integer a (:452)), 2 DUP (a, b = the two positions), 3 INV (aux: 0 "++", 1 "--";
 * a, b), 4 TRA (aux: 0..3 = 'A'..'D'; a = pos1, c = mate chromosome, b = pos2).  chr = the tuple's last element.
 */
typedef struct csv_split_in {
    int64_t         n_reads;
    const int64_t*  ent_off;        /* n_reads + 1 */
    const int64_t*  read_len;       /* read.query_length (total_L / RLength) */
    const int64_t*  c0;  const int64_t* c1;  const int64_t* f0;  const int64_t* f1;      /* per entry */
    const int32_t*  chr; const int32_t* mapq;
    const uint8_t*  strand;  const uint8_t* primary;
    int64_t         sv_size;        /* --min_size (parse_read's SV_size) */
    int64_t         max_size;       /* --max_size, -1: no limit */
    int32_t         min_mapq;
    int32_t         max_split_parts;   /* -1: no limit */
    /* CSV_CG_TO_POOL in `flags`: the candidates also become rows of the context's pool (csv_pool_* above).  A candidate of kind k
     * (0 DEL, 1 INS, 2 DUP, 3 INV, 4 TRA) on chromosome rank ch goes to segment pool_seg_base[k] + ch with the columns the rebuild
     * sorts on: (pos, len) for DEL / INS (an x.5 INS position as its integer part; aux = length of query[c:d], clipped to
     * query_len[read] - NULL: read_len - like a Python slice), (pos1, pos2) for DUP, aux = strand code for INV, aux = chr2 * 8 +
     * type for TRA; read index = read_base + the read's index in this batch.  Output arrays of csv_split_out that are NULL are
     * then not written. */
    int32_t         flags;
    int32_t         pool_seg_base[5];
    int64_t         read_base;
    const int32_t*  query_len;      /* n_reads or NULL */
} csv_split_in;

typedef struct csv_split_out {
    int64_t  cap;
    int64_t  n;                     /* out */
    uint8_t* kind;  int32_t* read;  int32_t* chr;  int32_t* aux;  int64_t* a;  int64_t* b;  int64_t* c;  int64_t* d;   /* cap */
    float    ms_device;             /* out: kernels only */
    int32_t  reserved;
} csv_split_out;

int csv_split_signatures(csv_ctx* ctx, const csv_split_in* in, csv_split_out* out);

/* ---------------------------------------------------------------------------------------------
 * Host-side VCF record emit: the structure-of-arrays result -> the text lines
 * of cuteSV's VCF body, without materialising Python row lists.  Replaces generate_output
 * (cuteSV_genotype.py:242-467: per-chromosome stable sort by POS, size filters, INFO/FORMAT
 * assembly, REF/ALT from the reference sequence, q5 filter, AF) and the SVID numbering of main_ctrl
 * (cuteSV main script :1208-1237: one counter per SV type over the sorted chromosome names).
 * No GPU work: plain C++ on the caller's thread.
 */
typedef struct csv_vcf_in {
    /* the calls (csv_batch_out after csv_cluster_batch / csv_batch_download) and their segments */
    const csv_batch_out* res;
    const csv_segment*   seg;
    int32_t              n_seg;
    int32_t              n_chrom;
    /* per chromosome (index = csv_segment.chrom): name, reference sequence, emission rank
     * (chrom_rank[c] = position of chromosome c in sorted(names): the order main_ctrl writes them) */
    const char* const*   chrom_name;
    const char* const*   chrom_seq;
    const int64_t*       chrom_len;
    const int32_t*       chrom_rank;
    /* strings attached to calls, as CSR blobs indexed by call (NULL = absent):
     *   ins_alt  the inserted sequence already sliced to SVLEN (INS calls; INDEL:402)
     *   rnames   comma-joined supporting read names (only read when report_readid)            */
    const char*          ins_alt;    const int64_t* ins_alt_off;
    const char*          rnames;     const int64_t* rnames_off;
    /* INV strand strings by call_aux code; genotype strings by table row:
     *   gl_key[n_gl] sorted gl_idx values present, gl_str = per key "GT\tPL\tGQ\tQUAL" (tab separated) */
    const char* const*   strand_name;
    const int32_t*       gl_key;     const char* const* gl_str;     int32_t n_gl;
    /* flags of cuteSV_Description.py: --min_size (:144), --max_size (:148), --genotype (:159), --report_readid (:100), --ignore_sequence (:104) */
    int64_t              min_size;
    int64_t              max_size;
    int32_t              genotype;
    int32_t              report_readid;
    int32_t              ignore_sequence;
    int32_t              reserved;
    /* (ABI v6) chrom_seq[c] may point INTO a FASTA file (mmap) instead of at a contiguous string: per chromosome the bases per
     * line and the bytes per line (bases + line break) of its `.fai` entry; NULL arrays or a 0 entry = contiguous */
    const int32_t*       chrom_line_bases;
    const int32_t*       chrom_line_width;
} csv_vcf_in;

/* Writes the records into `out` (capacity `cap` bytes) and returns CSV_OK, or CSV_E_CAPACITY with
 * *n_written = the needed size.  svid[5] = running record counters in the order INS, DEL, BND, DUP, INV
 * (main script :1209-1213); pass zeros for a fresh file, they are advanced in place. */
int csv_vcf_emit(const csv_vcf_in* in, char* out, int64_t cap, int64_t* n_written, int64_t* svid);

/* The `.fai` of a FASTA file held in memory (e.g. mmap): what pysam.FastaFile / `samtools faidx` give generate_output
 * (GT:254-259) - per contig its name (name_off / name_len address the header text inside `data`, up to the first white space),
 * length, byte offset of the first base, bases per line and bytes per line.  Returns the number of contigs found (call again
 * with larger arrays when it exceeds max_contigs; NULL arrays are skipped), or -CSV_E_INVALID for a file that cannot
 * be indexed (sequence before a header, lines of unequal length inside a contig).  No GPU work. */
int64_t csv_fasta_index(const char* data, int64_t size, int64_t max_contigs, int64_t* name_off, int32_t* name_len,
                        int64_t* length, int64_t* offset, int32_t* line_bases, int32_t* line_width);

/* ---------------------------------------------------------------------------------------------
 * Host-side row builder: the structure-of-arrays result -> the row lists the reference's resolvers return
 * (DEL 13 fields INDEL:207-219 / 464-478, INS 14 INDEL:419-432, DUP 11 DUP:121-131 / 170-180, INV 12 INV:145-156 /
 * 240-251, TRA 12 TRA:171-182), as ONE text blob: fields separated by '\t', rows terminated by '\n', rows in call
 * order (row c belongs to segment res->call_seg[c]).  All numeric fields are decimal text, read names joined by ','
 * exactly as the reference's rows hold them.  The Python shim splits the blob once (cutesv_amd/rows.py); any other
 * host language can do the same.  No GPU work: plain C++ on the caller's thread.
 */
typedef struct csv_rows_in {
    const csv_batch_out* res;
    const csv_segment*   seg;
    int32_t              n_seg;
    int32_t              n_chrom;
    const char* const*   chrom_name;     /* n_chrom */
    const int32_t*       read_id;        /* the batch's read_id column (support_sig indexes it) */
    const int32_t*       aux;            /* the batch's aux column (length of a synthetic inserted sequence) */
    /* read names: a table (blob + n_names + 1 offsets) or, when name_blob is NULL, "<name_prefix><id zero-padded to
     * name_width digits>" (the naming scheme of the synthetic workloads) */
    const char*          name_blob;
    const int64_t*       name_off;
    int64_t              n_names;
    const char*          name_prefix;
    int32_t              name_width;
    int32_t              n_strand;
    /* inserted sequences by global signature index (blob + n_sig + 1 offsets); NULL = "ACGT" repeated to aux[sig] */
    const char*          ins_blob;
    const int64_t*       ins_off;
    const char* const*   strand_name;    /* INV: call_aux -> strand text */
    /* cal_GL's strings per table row: CSV_GL_TABLE_SIZE entries "GT\tPL\tGQ\tQUAL" as blob + offsets */
    const char*          gl_blob;
    const int64_t*       gl_off;
} csv_rows_in;

/* Writes the rows into `out` (capacity `cap`) and returns CSV_OK, or CSV_E_CAPACITY with *n_written = the need. */
int csv_rows_emit(const csv_rows_in* in, char* out, int64_t cap, int64_t* n_written);

#ifdef __cplusplus
}
#endif
#endif /* CUTESV_HIP_H */
