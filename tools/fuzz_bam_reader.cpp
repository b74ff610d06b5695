#include "bam_reader.hpp"
#include <cstdio>
#include <cstdlib>
int main(int argc, char** argv) {
    for (int a = 1; a < argc; ++a) {
        for (int seek = -1; seek <= 1; seek += 2) {
            gdh::BamReader rd;
            std::string err;
            if (!rd.open(argv[a], 2, &err)) continue;
            if (seek >= 0) rd.seek_contig(seek, &err);
            std::vector<std::vector<uint64_t>> lin;
            gdh::BamReader::linear_index(argv[a], &lin, &err);
            gdh::RecordBlock blk;
            for (;;) { int rc = rd.next_block(blk, 100, &err); if (rc <= 0) break; }
        }
    }
    puts("done");
}
