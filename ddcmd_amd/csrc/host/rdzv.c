/*
 * rdzv.c -- process rendezvous and small host collectives over TCP.
 *
 * ddcMD gets its ranks, MPI_Bcast, MPI_Barrier and MPI_Allreduce from the MPI
 * launcher (ddcMD.c:93-139, energyInfo.c:9-63).  A one-process-per-GPU launch
 * without MPI (python -m torch.distributed.run, srun, a shell loop) only hands
 * every process RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.  This file turns
 * those into a full mesh of TCP streams with the handful of collectives the
 * Martini path needs around its RCCL data plane: the broadcast of the 128-byte
 * RCCL id, barriers, max/sum of a few doubles, and -- for tests and for nodes
 * whose GPUs cannot reach each other through RCCL -- a host-staged message
 * exchange with the matching rule of ncclSend/ncclRecv (per peer, in order).
 *
 * Bootstrap: rank 0 listens on (addr, port); every other rank opens its own
 * listening socket on an ephemeral port, connects to rank 0 and reports it;
 * rank 0 returns the table; rank i then connects to every rank 0 < j < i.  When
 * the launcher itself occupies MASTER_PORT (torch.distributed.run keeps its
 * store there) the port is passed as 0 together with a file name: rank 0 binds
 * an ephemeral port and publishes it in that file (written under a temporary
 * name and renamed), the others poll the file.
 */
#define _GNU_SOURCE
#include "ddcmi.h"
#include <arpa/inet.h>
#include <errno.h>
#include <fcntl.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <sys/types.h>
#include <time.h>
#include <unistd.h>

#define RDZV_MAGIC 0x64646372u      /* "ddcr" */

struct ddcmi_rdzv
{
   int rank, world;
   int *fd;                 /* fd[r]: stream to rank r; -1 for this rank */
   double timeout_s;
   char err[320];
};

static char g_rdzv_err[320];

static double now_s(void)
{
   struct timespec ts;
   clock_gettime(CLOCK_MONOTONIC, &ts);
   return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static int fail(ddcmi_rdzv *h, const char *fmt, ...)
{
   va_list ap;
   va_start(ap, fmt);
   vsnprintf(h ? h->err : g_rdzv_err, 320, fmt, ap);
   va_end(ap);
   return DDCMI_ECOMM;
}
static void tune(int fd)
{
   int one = 1;
   (void)setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
}
static int set_nonblock(int fd, int on)
{
   int fl = fcntl(fd, F_GETFL, 0);
   if (fl < 0) return -1;
   return fcntl(fd, F_SETFL, on ? (fl | O_NONBLOCK) : (fl & ~O_NONBLOCK));
}
/* blocking transfer of exactly n bytes with a deadline */
static int xfer(int fd, void *buf, size_t n, int out, double deadline)
{
   char *p = (char *)buf;
   while (n > 0)
   {
      double left = deadline - now_s();
      if (left <= 0) return -2;
      struct pollfd pf = {fd, (short)(out ? POLLOUT : POLLIN), 0};
      int pr = poll(&pf, 1, (int)(left * 1000.0) + 1);
      if (pr < 0) { if (errno == EINTR) continue; return -1; }
      if (pr == 0) return -2;
      ssize_t k = out ? send(fd, p, n, MSG_NOSIGNAL) : recv(fd, p, n, 0);
      if (k < 0) { if (errno == EINTR || errno == EAGAIN || errno == EWOULDBLOCK) continue; return -1; }
      if (k == 0 && !out) return -3;      /* peer closed */
      p += k; n -= (size_t)k;
   }
   return 0;
}
static int put(ddcmi_rdzv *h, int r, const void *buf, size_t n)
{
   int rc = xfer(h->fd[r], (void *)buf, n, 1, now_s() + h->timeout_s);
   if (rc) return fail(h, "rank %d: send of %zu bytes to rank %d failed (%s)", h->rank, n, r, rc == -2 ? "timeout" : strerror(errno));
   return 0;
}
static int get(ddcmi_rdzv *h, int r, void *buf, size_t n)
{
   int rc = xfer(h->fd[r], buf, n, 0, now_s() + h->timeout_s);
   if (rc) return fail(h, "rank %d: receive of %zu bytes from rank %d failed (%s)", h->rank, n, r, rc == -2 ? "timeout" : rc == -3 ? "peer closed the connection" : strerror(errno));
   return 0;
}

static int listen_on(const char *addr, int port, int *port_out)
{
   int fd = socket(AF_INET, SOCK_STREAM, 0);
   if (fd < 0) return -1;
   int one = 1;
   (void)setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
   struct sockaddr_in sa;
   memset(&sa, 0, sizeof(sa));
   sa.sin_family = AF_INET;
   sa.sin_port = htons((unsigned short)port);
   sa.sin_addr.s_addr = htonl(INADDR_ANY);
   if (addr && *addr && inet_pton(AF_INET, addr, &sa.sin_addr) != 1) sa.sin_addr.s_addr = htonl(INADDR_ANY);
   if (bind(fd, (struct sockaddr *)&sa, sizeof(sa)) != 0 || listen(fd, 128) != 0) { close(fd); return -1; }
   socklen_t sl = sizeof(sa);
   if (getsockname(fd, (struct sockaddr *)&sa, &sl) == 0 && port_out) *port_out = ntohs(sa.sin_port);
   return fd;
}
static int connect_to(uint32_t ip_be, int port, double deadline)
{
   for (;;)
   {
      int fd = socket(AF_INET, SOCK_STREAM, 0);
      if (fd < 0) return -1;
      struct sockaddr_in sa;
      memset(&sa, 0, sizeof(sa));
      sa.sin_family = AF_INET;
      sa.sin_port = htons((unsigned short)port);
      sa.sin_addr.s_addr = ip_be;
      if (connect(fd, (struct sockaddr *)&sa, sizeof(sa)) == 0) { tune(fd); return fd; }
      close(fd);
      if (now_s() > deadline) return -1;
      usleep(20000);
   }
}
static int accept_one(int lfd, uint32_t *ip_be, double deadline)
{
   for (;;)
   {
      double left = deadline - now_s();
      if (left <= 0) return -1;
      struct pollfd pf = {lfd, POLLIN, 0};
      int pr = poll(&pf, 1, (int)(left * 1000.0) + 1);
      if (pr < 0 && errno != EINTR) return -1;
      if (pr <= 0) continue;
      struct sockaddr_in sa;
      socklen_t sl = sizeof(sa);
      int fd = accept(lfd, (struct sockaddr *)&sa, &sl);
      if (fd < 0) { if (errno == EINTR || errno == EAGAIN) continue; return -1; }
      tune(fd);
      if (ip_be) *ip_be = sa.sin_addr.s_addr;
      return fd;
   }
}

const char *ddcmi_rdzv_last_error(const ddcmi_rdzv *h) { return h ? h->err : g_rdzv_err; }
int ddcmi_rdzv_rank(const ddcmi_rdzv *h) { return h ? h->rank : -1; }
int ddcmi_rdzv_world(const ddcmi_rdzv *h) { return h ? h->world : 0; }

/* leave the job at once: every stream is shut down, so the peers' pending and next transfers fail instead of waiting */
void ddcmi_rdzv_abort(ddcmi_rdzv *h)
{
   if (!h || !h->fd) return;
   for (int r = 0; r < h->world; r++) if (h->fd[r] >= 0) { (void)shutdown(h->fd[r], SHUT_RDWR); close(h->fd[r]); h->fd[r] = -1; }
}
void ddcmi_rdzv_destroy(ddcmi_rdzv *h)
{
   if (!h) return;
   if (h->fd)
   {
      for (int r = 0; r < h->world; r++) if (h->fd[r] >= 0) close(h->fd[r]);
      free(h->fd);
   }
   free(h);
}

int ddcmi_rdzv_create(ddcmi_rdzv **out, int rank, int world, const char *addr, int port, const char *port_file, double timeout_s)
{
   if (!out || world < 1 || rank < 0 || rank >= world) return fail(NULL, "ddcmi_rdzv_create: bad rank %d of %d", rank, world), DDCMI_EINVAL;
   if (port <= 0 && (!port_file || !*port_file) && world > 1) return fail(NULL, "ddcmi_rdzv_create: neither a port nor a port file"), DDCMI_EINVAL;
   *out = NULL;
   ddcmi_rdzv *h = (ddcmi_rdzv *)calloc(1, sizeof(*h));
   if (!h) return DDCMI_ENOMEM;
   h->rank = rank; h->world = world; h->timeout_s = timeout_s > 0 ? timeout_s : 300.0;
   h->fd = (int *)malloc(sizeof(int) * (size_t)world);
   if (!h->fd) { free(h); return DDCMI_ENOMEM; }
   for (int r = 0; r < world; r++) h->fd[r] = -1;
   if (world == 1) { *out = h; return DDCMI_OK; }
   const double deadline = now_s() + h->timeout_s;
   uint32_t *ip = (uint32_t *)calloc((size_t)world, sizeof(uint32_t));
   int *lport = (int *)calloc((size_t)world, sizeof(int));
   int lfd = -1, rc = DDCMI_ECOMM;
   if (!ip || !lport) { rc = DDCMI_ENOMEM; goto done; }
   const int use_file = (port <= 0) || (port_file && *port_file);
   uint32_t ip0 = htonl(INADDR_LOOPBACK);
   if (addr && *addr) { struct in_addr ia; if (inet_pton(AF_INET, addr, &ia) == 1) ip0 = ia.s_addr; }
   if (use_file && ip0 != htonl(INADDR_LOOPBACK))
   {
      /* the port travels through a file, which the ranks of ONE node share: an address this host cannot bind is another node's */
      int pfd = listen_on(addr, 0, NULL);
      if (pfd < 0 && errno == EADDRNOTAVAIL)
      {
         fail(NULL, "rank %d: MASTER_ADDR %s is not an address of this host, and the rendezvous port travels through a file (%s) that only the ranks of one node see: set DDCMI_RDZV_PORT to a free port for launches over several nodes",
              rank, addr, port_file && *port_file ? port_file : "?");
         goto done;
      }
      if (pfd >= 0) close(pfd);
   }
   if (rank == 0)
   {
      int myport = 0;
      /* on the address the ranks were given (loopback for 127.0.0.1), not on every interface: nobody else reaches the port */
      lfd = listen_on(addr && *addr ? addr : "127.0.0.1", use_file ? 0 : port, &myport);
      if (lfd < 0) lfd = listen_on(NULL, use_file ? 0 : port, &myport);      /* MASTER_ADDR is a name or not an address of this host: any interface */
      if (lfd < 0) { fail(NULL, "rank 0: cannot listen on port %d: %s", port, strerror(errno)); goto done; }
      if (use_file)
      {
         char tmp[1100];
         snprintf(tmp, sizeof(tmp), "%s.%d.tmp", port_file, (int)getpid());
         /* a fresh file of our own: never through a link somebody planted under the predictable name */
         (void)unlink(tmp);
         int tfd = open(tmp, O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);
         FILE *f = tfd >= 0 ? fdopen(tfd, "w") : NULL;
         if (!f) { if (tfd >= 0) close(tfd); fail(NULL, "rank 0: cannot write %s: %s", tmp, strerror(errno)); goto done; }
         fprintf(f, "%d %d\n", myport, (int)getpid());
         fclose(f);
         if (rename(tmp, port_file) != 0) { fail(NULL, "rank 0: rename to %s failed: %s", port_file, strerror(errno)); unlink(tmp); goto done; }
      }
      for (int k = 1; k < world; k++)
      {
         uint32_t pip = 0;
         int fd = accept_one(lfd, &pip, deadline);
         if (fd < 0) { fail(NULL, "rank 0: only %d of %d ranks arrived within %.0f s", k, world, h->timeout_s); goto done; }
         uint32_t hello[4];
         /* the peer's numbers are compared as the unsigned values they index with; a connection that stays silent
          * for two seconds is dropped instead of holding the accept loop up to the global deadline */
         const double hello_by = now_s() + 2.0 < deadline ? now_s() + 2.0 : deadline;
         if (xfer(fd, hello, sizeof(hello), 0, hello_by) || hello[0] != RDZV_MAGIC || hello[2] != (uint32_t)world || hello[1] == 0 || hello[1] >= (uint32_t)world ||
             hello[3] == 0 || hello[3] > 65535u || h->fd[hello[1]] >= 0)
         { close(fd); k--; if (now_s() > deadline) { fail(NULL, "rank 0: rendezvous timed out"); goto done; } continue; }      /* a stray connection: ignore it */
         h->fd[hello[1]] = fd; ip[hello[1]] = pip; lport[hello[1]] = (int)hello[3];
      }
      if (use_file) unlink(port_file);
      for (int r = 1; r < world; r++)
         if (xfer(h->fd[r], ip, sizeof(uint32_t) * (size_t)world, 1, deadline) || xfer(h->fd[r], lport, sizeof(int) * (size_t)world, 1, deadline))
         { fail(NULL, "rank 0: sending the address table to rank %d failed", r); goto done; }
   }
   else
   {
      int myport = 0;
      lfd = listen_on(NULL, 0, &myport);
      if (lfd < 0) { fail(NULL, "rank %d: cannot open a listening socket: %s", rank, strerror(errno)); goto done; }
      for (;;)
      {
         int p0 = port;
         if (use_file)
         {
            p0 = 0;
            /* the file must be a plain file of this user: a planted one would send the ranks (and the RCCL id) elsewhere */
            int pfd = open(port_file, O_RDONLY | O_NOFOLLOW);
            struct stat stf;
            FILE *f = NULL;
            if (pfd >= 0 && fstat(pfd, &stf) == 0 && S_ISREG(stf.st_mode) && stf.st_uid == geteuid()) f = fdopen(pfd, "r");
            else if (pfd >= 0) close(pfd);
            if (f) { int pid = 0; if (fscanf(f, "%d %d", &p0, &pid) < 1 || p0 > 65535) p0 = 0; fclose(f); }
         }
         if (p0 > 0)
         {
            int fd = connect_to(ip0, p0, now_s() + 0.5);
            if (fd >= 0)
            {
               uint32_t hello[4] = {RDZV_MAGIC, (uint32_t)rank, (uint32_t)world, (uint32_t)myport};
               if (xfer(fd, hello, sizeof(hello), 1, deadline) == 0 && xfer(fd, ip, sizeof(uint32_t) * (size_t)world, 0, deadline) == 0 &&
                   xfer(fd, lport, sizeof(int) * (size_t)world, 0, deadline) == 0) { h->fd[0] = fd; break; }
               close(fd);      /* a stale port file of an earlier run, or rank 0 gave up */
            }
         }
         if (now_s() > deadline) { fail(NULL, "rank %d: rank 0 did not answer at %s:%d%s%s within %.0f s", rank, addr ? addr : "127.0.0.1", p0, use_file ? " from " : "", use_file ? port_file : "", h->timeout_s); goto done; }
         usleep(50000);
      }
      /* connections between the other ranks: the higher rank connects, the lower accepts */
      for (int j = 1; j < rank; j++)
      {
         uint32_t pip = ip[j] ? ip[j] : ip0;
         int fd = connect_to(pip, lport[j], deadline);
         if (fd < 0) { fail(NULL, "rank %d: cannot reach rank %d", rank, j); goto done; }
         uint32_t hello[2] = {RDZV_MAGIC, (uint32_t)rank};
         if (xfer(fd, hello, sizeof(hello), 1, deadline)) { close(fd); fail(NULL, "rank %d: hello to rank %d failed", rank, j); goto done; }
         h->fd[j] = fd;
      }
      for (int k = rank + 1; k < world; k++)
      {
         int fd = accept_one(lfd, NULL, deadline);
         if (fd < 0) { fail(NULL, "rank %d: ranks above did not connect", rank); goto done; }
         uint32_t hello[2];
         const double hello_by = now_s() + 2.0 < deadline ? now_s() + 2.0 : deadline;
         if (xfer(fd, hello, sizeof(hello), 0, hello_by) || hello[0] != RDZV_MAGIC || hello[1] <= (uint32_t)rank || hello[1] >= (uint32_t)world || h->fd[hello[1]] >= 0)
         { close(fd); k--; if (now_s() > deadline) { fail(NULL, "rank %d: ranks above did not connect", rank); goto done; } continue; }
         h->fd[hello[1]] = fd;
      }
   }
   rc = DDCMI_OK;
done:
   if (lfd >= 0) close(lfd);
   free(ip); free(lport);
   if (rc != DDCMI_OK) { ddcmi_rdzv_destroy(h); return rc; }
   *out = h;
   return ddcmi_rdzv_barrier(h);
}

/* MPI_Bcast */
int ddcmi_rdzv_bcast(ddcmi_rdzv *h, void *buf, size_t nbytes, int root)
{
   if (!h || !buf || root < 0 || root >= h->world) return DDCMI_EINVAL;
   if (h->world == 1 || nbytes == 0) return DDCMI_OK;
   int rc;
   if (root != 0)
   {
      if (h->rank == root && (rc = put(h, 0, buf, nbytes))) return rc;
      if (h->rank == 0 && (rc = get(h, root, buf, nbytes))) return rc;
   }
   if (h->rank == 0) { for (int r = 1; r < h->world; r++) if (r != root && (rc = put(h, r, buf, nbytes))) return rc; }
   else if (h->rank != root && (rc = get(h, 0, buf, nbytes))) return rc;
   return DDCMI_OK;
}
/* MPI_Barrier */
int ddcmi_rdzv_barrier(ddcmi_rdzv *h)
{
   if (!h) return DDCMI_EINVAL;
   if (h->world == 1) return DDCMI_OK;
   char c = 1;
   int rc;
   if (h->rank == 0)
   {
      for (int r = 1; r < h->world; r++) if ((rc = get(h, r, &c, 1))) return rc;
      for (int r = 1; r < h->world; r++) if ((rc = put(h, r, &c, 1))) return rc;
   }
   else { if ((rc = put(h, 0, &c, 1))) return rc; if ((rc = get(h, 0, &c, 1))) return rc; }
   return DDCMI_OK;
}
/* MPI_Allreduce of doubles, op 0 = sum (in rank order: the same bits on every rank, every run), 1 = max */
int ddcmi_rdzv_allreduce_f64(ddcmi_rdzv *h, double *v, int n, int op)
{
   if (!h || !v || n <= 0 || op < 0 || op > 1) return DDCMI_EINVAL;
   if (h->world == 1) return DDCMI_OK;
   int rc;
   if (h->rank == 0)
   {
      double *t = (double *)malloc(sizeof(double) * (size_t)n);
      if (!t) return DDCMI_ENOMEM;
      for (int r = 1; r < h->world; r++)
      {
         if ((rc = get(h, r, t, sizeof(double) * (size_t)n))) { free(t); return rc; }
         for (int k = 0; k < n; k++) v[k] = op ? (t[k] > v[k] ? t[k] : v[k]) : v[k] + t[k];
      }
      free(t);
      for (int r = 1; r < h->world; r++) if ((rc = put(h, r, v, sizeof(double) * (size_t)n))) return rc;
   }
   else
   {
      if ((rc = put(h, 0, v, sizeof(double) * (size_t)n))) return rc;
      if ((rc = get(h, 0, v, sizeof(double) * (size_t)n))) return rc;
   }
   return DDCMI_OK;
}
/* MPI_Allgather of nbytes per rank */
int ddcmi_rdzv_allgather(ddcmi_rdzv *h, const void *send, void *recv, size_t nbytes)
{
   if (!h || !send || !recv) return DDCMI_EINVAL;
   char *all = (char *)recv;
   memmove(all + (size_t)h->rank * nbytes, send, nbytes);
   if (h->world == 1 || nbytes == 0) return DDCMI_OK;
   int rc;
   if (h->rank == 0)
   {
      for (int r = 1; r < h->world; r++) if ((rc = get(h, r, all + (size_t)r * nbytes, nbytes))) return rc;
      for (int r = 1; r < h->world; r++) if ((rc = put(h, r, all, nbytes * (size_t)h->world))) return rc;
   }
   else
   {
      if ((rc = put(h, 0, all + (size_t)h->rank * nbytes, nbytes))) return rc;
      if ((rc = get(h, 0, all, nbytes * (size_t)h->world))) return rc;
   }
   return DDCMI_OK;
}

/* Grouped point-to-point exchange with the matching rule of ncclSend/ncclRecv inside one
 * ncclGroupStart/End: the k-th message this rank sends to peer p is the k-th message p
 * receives from this rank.  All streams progress together (poll), so neither the order of the
 * peers nor the message sizes can deadlock.  A message to the rank itself is copied. */
int ddcmi_rdzv_exchange(ddcmi_rdzv *h, int nsend, const int *send_peer, const void *const *send_buf, const size_t *send_bytes,
                        int nrecv, const int *recv_peer, void *const *recv_buf, const size_t *recv_bytes)
{
   if (!h || nsend < 0 || nrecv < 0) return DDCMI_EINVAL;
   for (int k = 0; k < nsend; k++) if (send_peer[k] < 0 || send_peer[k] >= h->world) return fail(h, "exchange: send peer %d out of range", send_peer[k]), DDCMI_EINVAL;
   for (int k = 0; k < nrecv; k++) if (recv_peer[k] < 0 || recv_peer[k] >= h->world) return fail(h, "exchange: receive peer %d out of range", recv_peer[k]), DDCMI_EINVAL;
   /* self messages, in order */
   {
      int r = 0;
      for (int s = 0; s < nsend; s++)
      {
         if (send_peer[s] != h->rank) continue;
         while (r < nrecv && recv_peer[r] != h->rank) r++;
         if (r == nrecv || recv_bytes[r] != send_bytes[s]) return fail(h, "exchange: message to self without a matching receive"), DDCMI_EINVAL;
         memcpy(recv_buf[r], send_buf[s], send_bytes[s]);
         r++;
      }
   }
   const int W = h->world;
   int *scur = (int *)malloc(sizeof(int) * 2 * (size_t)W), *rcur = scur ? scur + W : NULL;      /* per peer: index of the message in progress */
   size_t *sdone = (size_t *)calloc(2 * (size_t)W, sizeof(size_t)), *rdone = sdone ? sdone + W : NULL;
   struct pollfd *pf = (struct pollfd *)malloc(sizeof(struct pollfd) * (size_t)W);
   if (!scur || !sdone || !pf) { free(scur); free(sdone); free(pf); return DDCMI_ENOMEM; }
   int rc = DDCMI_OK;
#define NEXT_MSG(cur, n, peerarr, bytes, p) do { while ((cur)[p] < (n) && ((peerarr)[(cur)[p]] != (p) || (bytes)[(cur)[p]] == 0)) (cur)[p]++; } while (0)
   for (int p = 0; p < W; p++)
   {
      scur[p] = rcur[p] = 0;
      if (p == h->rank) { scur[p] = nsend; rcur[p] = nrecv; continue; }
      NEXT_MSG(scur, nsend, send_peer, send_bytes, p);
      NEXT_MSG(rcur, nrecv, recv_peer, recv_bytes, p);
      if (h->fd[p] >= 0) (void)set_nonblock(h->fd[p], 1);
   }
   const double deadline = now_s() + h->timeout_s;
   for (;;)
   {
      int npf = 0;
      for (int p = 0; p < W; p++)
      {
         short ev = 0;
         if (scur[p] < nsend) ev |= POLLOUT;
         if (rcur[p] < nrecv) ev |= POLLIN;
         if (ev) { pf[npf].fd = h->fd[p]; pf[npf].events = ev; pf[npf].revents = 0; npf++; }
      }
      if (npf == 0) break;
      double left = deadline - now_s();
      if (left <= 0)
      {
         /* say WHOSE data never came (and who never took ours): the peers with a message still in progress */
         char who[192];
         int o = 0;
         who[0] = 0;
         for (int p = 0; p < W && o < (int)sizeof(who) - 40; p++)
            if (scur[p] < nsend || rcur[p] < nrecv)
               o += snprintf(who + o, sizeof(who) - (size_t)o, "%s rank %d (%s%s%s)", o ? "," : "", p, rcur[p] < nrecv ? "nothing received" : "", (rcur[p] < nrecv && scur[p] < nsend) ? ", " : "", scur[p] < nsend ? "send not taken" : "");
         rc = fail(h, "rank %d: message exchange timed out after %.0f s waiting for%s", h->rank, h->timeout_s, who);
         break;
      }
      int pr = poll(pf, (nfds_t)npf, (int)(left * 1000.0) + 1);
      if (pr < 0) { if (errno == EINTR) continue; rc = fail(h, "poll: %s", strerror(errno)); break; }
      int q = 0;
      for (int p = 0; p < W && rc == DDCMI_OK; p++)
      {
         if (!(scur[p] < nsend || rcur[p] < nrecv)) continue;
         const short re = pf[q++].revents;
         if ((re & (POLLOUT | POLLERR | POLLHUP)) && scur[p] < nsend)
         {
            const int m = scur[p];
            ssize_t k = send(h->fd[p], (const char *)send_buf[m] + sdone[p], send_bytes[m] - sdone[p], MSG_NOSIGNAL);
            if (k < 0 && errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR) { rc = fail(h, "rank %d: send to rank %d: %s", h->rank, p, strerror(errno)); break; }
            if (k > 0) sdone[p] += (size_t)k;
            if (sdone[p] == send_bytes[m]) { sdone[p] = 0; scur[p]++; NEXT_MSG(scur, nsend, send_peer, send_bytes, p); }
         }
         if ((re & (POLLIN | POLLERR | POLLHUP)) && rcur[p] < nrecv)
         {
            const int m = rcur[p];
            ssize_t k = recv(h->fd[p], (char *)recv_buf[m] + rdone[p], recv_bytes[m] - rdone[p], 0);
            if (k == 0) { rc = fail(h, "rank %d: rank %d closed the connection during an exchange", h->rank, p); break; }
            if (k < 0 && errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR) { rc = fail(h, "rank %d: receive from rank %d: %s", h->rank, p, strerror(errno)); break; }
            if (k > 0) rdone[p] += (size_t)k;
            if (rdone[p] == recv_bytes[m]) { rdone[p] = 0; rcur[p]++; NEXT_MSG(rcur, nrecv, recv_peer, recv_bytes, p); }
         }
      }
      if (rc != DDCMI_OK) break;
   }
#undef NEXT_MSG
   for (int p = 0; p < W; p++) if (p != h->rank && h->fd[p] >= 0) (void)set_nonblock(h->fd[p], 0);
   free(scur); free(sdone); free(pf);
   return rc;
}
