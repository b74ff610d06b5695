// goleft-depth: the `depth` entry of the reference's dispatcher
// (/root/reference/cmd/goleft/goleft.go:25-32,:55-69) as a C++ executable.
//   goleft-depth [flags] BAM          or          goleft-depth depth [flags] BAM
#include <cstring>
#include <vector>

#include "../../../include/goleft_depth_host.h"

int main(int argc, char** argv)
{
    std::vector<const char*> av;
    av.push_back("goleft depth");
    int first = 1;
    if (argc > 1 && strcmp(argv[1], "depth") == 0) first = 2;
    for (int i = first; i < argc; ++i) av.push_back(argv[i]);
    return gdh_depth_main((int)av.size(), av.data());
}
