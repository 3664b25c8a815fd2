/* omnitok_comm.h -- the path's one collective as a C ABI: an RCCL all-gather of the token ids, issued on a caller-given
 * HIP stream from C++ (last step: "a direct ncclAllGather call on the op stream from C++").
 *
 * Reference launch contract: /root/reference/ddp_utils.py:333-364 (one process per GPU, RANK / WORLD_SIZE / LOCAL_RANK /
 * MASTER_ADDR / MASTER_PORT in the environment, init_process_group("nccl")).  The reference's encode()/decode() perform
 * no communication at all (OmniTokenizer/omnitokenizer.py:247-317); the id gather is what a clip-sharded caller adds.
 *
 * librccl.so is resolved at run time (dlopen: first the copy already mapped into the process -- PyTorch-ROCm's --, then
 * the system one), so libomnitok.so carries no link-time dependency on it and loads on a box without RCCL; every entry
 * point below returns OMNITOK_ERR_UNSUPPORTED with a message when the library cannot be found.
 *
 * Bootstrap: rank 0 calls omnitok_comm_unique_id and ships the 128 bytes to the other ranks by whatever channel the
 * caller already has (omnitokenizer_amd/dist.py: a broadcast on the existing torch.distributed process group; a bare C
 * caller: a file, a socket, MPI); every rank then calls omnitok_comm_create with the same bytes.
 */
#ifndef OMNITOK_COMM_H
#define OMNITOK_COMM_H
#include "omnitok.h"
#ifdef __cplusplus
extern "C" {
#endif

#define OMNITOK_COMM_ID_BYTES 128 /* sizeof(ncclUniqueId) */

typedef struct omnitok_comm omnitok_comm;

/* 1 when librccl.so could be resolved (and which one: path-ish description in `where`, may be NULL), else 0 */
int omnitok_comm_available(char *where, int where_len);
/* rank 0: fills id[OMNITOK_COMM_ID_BYTES] (ncclGetUniqueId) */
int omnitok_comm_unique_id(unsigned char *id);
/* every rank, collectively: ncclCommInitRank on the CURRENT device */
int omnitok_comm_create(const unsigned char *id, int rank, int world, omnitok_comm **out);
void omnitok_comm_destroy(omnitok_comm *c);
int omnitok_comm_world(omnitok_comm *c);
int omnitok_comm_rank(omnitok_comm *c);
/* recv[world * count] <- send[count] of every rank, int32 elements, rank order; stream-ordered (no host wait).
 * In-place (send == recv + rank * count) is allowed, as in NCCL. */
int omnitok_comm_allgather_i32(omnitok_comm *c, const int32_t *send, int32_t *recv, int64_t count,
                               omnitok_stream_t stream);
/* The path's gather in one stream-ordered call: narrows ids_local [count] int64 -> int32 into the communicator's send
 * block, all-gathers, widens into ids_all [world * count] int64.  Staging blocks are owned by the communicator and grow
 * on demand (hipMalloc; call once before capturing a graph).  ids < 2^31 (n_codes <= 32768 on this path). */
int omnitok_comm_allgather_ids(omnitok_comm *c, const int64_t *ids_local, int64_t *ids_all, int64_t count,
                               omnitok_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
