// bamwalk_emul.cpp -- TEST INFRASTRUCTURE (tests/test_bamwalk_emul.py): the device BAM record walk
// (goleft_amd/csrc/gd_bamdecode.hpp) compiled for the host and run lane by lane, with the inflated stream and every
// output array ending exactly at an inaccessible page -- a walk that trusts a damaged length field one byte too far
// ends the process.  On the GPU such a read is silent; here it is a failed test.
//   clang++ -O2 -std=c++17 -shared -fPIC -o bamwalk_emul.so bamwalk_emul.cpp
#include "emul_machine.hpp"
#include "../../goleft_amd/csrc/gd_bamdecode.hpp"

static const gd::BamSegJob* g_job = nullptr;
static void body_count() { gd::gd_bam_walk_kernel<false>(*g_job); }
static void body_extract() { gd::gd_bam_walk_kernel<true>(*g_job); }
static void body_extract_tab() { gd::gd_bam_extract_tab_kernel(*g_job); }
// 1 (the product's way): the counting walk leaves the record table, gd_bam_extract_tab_kernel reads it; 0: the second walk
static int g_tab = 1;
extern "C" void emul_bam_walk_mode(int tab) { g_tab = tab; }

// The two passes as gd_api_ingest.inc runs them: count, prefix sums on the host, extract.  Outputs: per segment
// n_rec / n_ops / first / last / flags [n_seg]; *n_records / *n_ops_total; and, when every flag is clean of bits 1, 2
// and no count overflows the caller's capacities, the arrays (cigar_off has n_records entries).
extern "C" int emul_bam_walk(const uint8_t* data, uint64_t n_bytes, const uint64_t* seg_beg, const uint64_t* seg_end, uint32_t n_seg,
                             int32_t tid, int32_t n_ref, uint32_t* n_rec, uint64_t* n_ops, int32_t* first, int32_t* last,
                             uint32_t* flags, uint64_t* n_records, uint64_t* n_ops_total, uint64_t cap_rec, uint64_t cap_ops,
                             int32_t* pos, uint16_t* flag, uint8_t* mapq, uint32_t* cigar_off, uint32_t* cigar)
{
    emul::Guarded d(n_bytes ? n_bytes : 1);
    memcpy(d.p, data, n_bytes);
    gd::BamSegJob j{};
    j.data = d.p; j.n_bytes = n_bytes; j.seg_beg = seg_beg; j.seg_end = seg_end; j.tid = tid; j.n_ref = n_ref; j.n_seg = n_seg;
    j.n_rec = n_rec; j.n_ops = n_ops; j.first_pos = first; j.last_pos = last; j.flags = flags;
    // the record table: every segment's share exactly (seg_end - seg_beg) / 36 + 1 entries, the whole behind a guard page
    std::vector<uint64_t> tbase(n_seg);
    uint64_t n_tab = 0;
    for (uint32_t s = 0; s < n_seg; ++s) { tbase[s] = n_tab; n_tab += (seg_end[s] > seg_beg[s] ? (seg_end[s] - seg_beg[s]) / 36 : 0) + 1; }
    emul::Guarded g_tabmem(n_tab * 8 + 8);
    if (g_tab) { j.tab = reinterpret_cast<uint32_t*>(g_tabmem.p + 8); j.tab_base = tbase.data(); }
    g_job = &j;
    for (unsigned b = 0; b < n_seg; ++b) emul::run(body_count, 64, b);          // one wave per segment
    std::vector<uint64_t> rbase(n_seg), obase(n_seg);
    uint64_t N = 0, M = 0;
    bool clean = true;
    for (uint32_t s = 0; s < n_seg; ++s) {
        rbase[s] = N; obase[s] = M;
        N += n_rec[s]; M += n_ops[s];
        if (flags[s] & 6u) clean = false;
    }
    *n_records = N; *n_ops_total = M;
    if (!clean || N > cap_rec || M > cap_ops) return 1;
    // every output exactly as large as the count pass said, each behind its own guard page
    emul::Guarded g_pos(N * 4 + 4), g_flag(N * 2 + 2), g_mapq(N + 1), g_off(N * 4 + 4), g_cig(M * 4 + 4);
    j.rec_base = rbase.data(); j.op_base = obase.data();
    j.pos = reinterpret_cast<int32_t*>(g_pos.p + 4); j.flag = reinterpret_cast<uint16_t*>(g_flag.p + 2); j.mapq = g_mapq.p + 1;
    j.cigar_off = reinterpret_cast<uint32_t*>(g_off.p + 4); j.cigar = reinterpret_cast<uint32_t*>(g_cig.p + 4);
    // (gd_bam_extract_tab_kernel strides by its block size: 256 on the device, one wave's worth of fibers here)
    if (g_tab) { blockDim.x = 64; for (unsigned b = 0; b < n_seg; ++b) emul::run(body_extract_tab, 64, b); }
    else { for (unsigned b = 0; b < n_seg; ++b) emul::run(body_extract, 64, b); }
    memcpy(pos, j.pos, N * 4); memcpy(flag, j.flag, N * 2); memcpy(mapq, j.mapq, N);
    memcpy(cigar_off, j.cigar_off, N * 4); memcpy(cigar, j.cigar, M * 4);
    return 0;
}
