/* TEST INFRASTRUCTURE (oracle/): path redirect for the compiled reference.
 *
 * The reference hard-codes its data directories as class members
 * (mixed_precs_caching/evlfu_8.hpp:58, evlfu_4.hpp:61, evlfu_16.hpp:64,
 * evlfu_32.hpp:61, aprx_embedding.hpp:39), all under /mnt/extra/ev-store-dlrm/.
 * Linked into the same shared object / binary as the reference sources, this
 * fopen() rewrites that prefix to $EVS_REF_ROOT so synthetic tables can live
 * inside the repo (or a temp dir) without touching the reference sources.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static const char kPrefix[] = "/mnt/extra/ev-store-dlrm/";

FILE *fopen(const char *path, const char *mode) {
    static FILE *(*real_fopen)(const char *, const char *) = NULL;
    if (!real_fopen) real_fopen = (FILE * (*)(const char *, const char *)) dlsym(RTLD_NEXT, "fopen");
    const char *root = getenv("EVS_REF_ROOT");
    if (root && path && strncmp(path, kPrefix, sizeof(kPrefix) - 1) == 0) {
        char buf[4096];
        snprintf(buf, sizeof buf, "%s/%s", root, path + sizeof(kPrefix) - 1);
        return real_fopen(buf, mode);
    }
    return real_fopen(path, mode);
}
