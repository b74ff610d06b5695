// goleft-depth: the `depth` entry of the reference's dispatcher
// (/root/reference/cmd/goleft/goleft.go:25-32,:55-69) as a C++ executable.
//   goleft-depth [flags] BAM          or          goleft-depth depth [flags] BAM
//   goleft-depth depthwed -s SIZE a.depth.bed b.depth.bed ...   (the consumer of depth.bed files)
//   goleft-depth multidepth -c CHROM a.bam b.bam ...            (/root/reference/multidepth, its own binary there)
//   samtools depth -Q q -d D -r REGION in.bam                   (the same program installed under the NAME samtools,
//                                                                goleft_amd/shim/samtools: an unmodified goleft finds it on PATH)
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../../include/goleft_depth_host.h"

// The outputs are written and closed inside the *_main functions; what is left when they return is giving tens of
// gigabytes of HBM back page by page and unloading the HIP runtime -- work the kernel does for a dying process
// anyway, in a fraction of the time.  A command-line tool leaves at once.
static int leave(int rc)
{
    fflush(stdout);
    fflush(stderr);
    if (getenv("GOLEFT_SLOW_EXIT")) exit(rc);             // a profiler writes its report from an exit handler
    _exit(rc);
}

int main(int argc, char** argv)
{
    gdh_set_fast_exit(1);
    std::vector<const char*> av;
    {
        const char* base = strrchr(argv[0], '/');
        base = base ? base + 1 : argv[0];
        if (strcmp(base, "samtools") == 0) {
            for (int i = 0; i < argc; ++i) av.push_back(argv[i]);
            return leave(gdh_samtools_main((int)av.size(), av.data()));
        }
    }
    if (argc > 1 && strcmp(argv[1], "depthwed") == 0) {
        av.push_back("goleft depthwed");
        for (int i = 2; i < argc; ++i) av.push_back(argv[i]);
        return leave(gdh_depthwed_main((int)av.size(), av.data()));
    }
    if (argc > 1 && strcmp(argv[1], "multidepth") == 0) {
        av.push_back("multidepth");
        for (int i = 2; i < argc; ++i) av.push_back(argv[i]);
        return leave(gdh_multidepth_main((int)av.size(), av.data()));
    }
    av.push_back("goleft depth");
    int first = 1;
    if (argc > 1 && strcmp(argv[1], "depth") == 0) first = 2;
    for (int i = first; i < argc; ++i) av.push_back(argv[i]);
    return leave(gdh_depth_main((int)av.size(), av.data()));
}
