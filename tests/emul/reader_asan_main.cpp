// reader_asan_main.cpp -- TEST INFRASTRUCTURE (tests/test_host_reader_asan.py): the host BAM reader
// (goleft_amd/csrc/host/bam_reader.cpp) built with -fsanitize=address,undefined and run over damaged files -- every file
// given, from its start and after a seek to reference 1, in blocks of 100 records.  An out-of-bounds read that a plain
// build survives by luck ends this process with a report.
#include "bam_reader.hpp"
#include <cstdio>
int main(int argc, char** argv) {
    for (int a = 1; a < argc; ++a) {
        for (int seek = 0; seek < 2; ++seek) {
            gdh::BamReader r; std::string err;
            if (!r.open(argv[a], 3, &err)) { continue; }
            if (seek) r.seek_contig(1, &err);
            gdh::RecordBlock b; size_t n = 0;
            for (;;) { int rc = r.next_block(b, 100, &err); if (rc <= 0) break; n += b.size(); }
        }
    }
    printf("ok\n");
}
