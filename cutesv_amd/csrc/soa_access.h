// soa_access.h — reading a csv_batch_out as its producer filled it (ABI v7): coordinates are int64 or, with
// CSV_OUT_COORD_I32, int32 arrays behind the same pointers; optional fields may be NULL.  Used by the host-side consumers
// of the result (rows_layout.h, vcf_emit.cpp); no counterpart in the reference, whose rows are Python lists.
#pragma once
#include <stdint.h>

#include "../../include/cutesv_hip.h"

namespace csv_soa {
inline int64_t coord(const csv_batch_out& R, const int64_t* p, int64_t c)
{
    return (R.flags & CSV_OUT_COORD_I32) ? (int64_t)((const int32_t*)p)[c] : p[c];
}
inline int64_t bp1(const csv_batch_out& R, int64_t c) { return coord(R, R.bp1, c); }
inline int64_t bp2(const csv_batch_out& R, int64_t c) { return coord(R, R.bp2, c); }
inline int64_t seq_pick(const csv_batch_out& R, int64_t c) { return R.seq_pick ? coord(R, R.seq_pick, c) : -1; }
inline int32_t gl_idx(const csv_batch_out& R, int64_t c) { return R.gl_idx ? R.gl_idx[c] : -1; }
inline int32_t dr(const csv_batch_out& R, int64_t c) { return R.dr ? R.dr[c] : -1; }
inline int32_t cipos(const csv_batch_out& R, int64_t c) { return R.cipos ? R.cipos[c] : 0; }
inline int32_t cilen(const csv_batch_out& R, int64_t c) { return R.cilen ? R.cilen[c] : 0; }
inline int32_t call_aux(const csv_batch_out& R, int64_t c) { return R.call_aux ? R.call_aux[c] : 0; }
inline bool has_support_list(const csv_batch_out& R)
{
    return !(R.flags & CSV_OUT_NO_SUPPORT_LIST) && R.support_off && (R.support_sig || R.support_sig32);
}
}  // namespace csv_soa
