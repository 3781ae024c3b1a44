/* deck.c -- see deck.h. */
#include "deck.h"
#include "object.h"
#include "units.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <ctype.h>
#include <unistd.h>
#include <inttypes.h>

#define FAIL(...) do { snprintf(err, errlen, __VA_ARGS__); goto fail; } while (0)

int ddcmi_setup_sizeof(void) { return (int)sizeof(ddcmi_setup); }

static char *path_join(const char *dir, const char *name)
{
   if (name[0] == '/' || dir[0] == 0) return strdup(name);
   size_t a = strlen(dir), b = strlen(name);
   char *r = malloc(a + b + 2);
   memcpy(r, dir, a); r[a] = '/'; memcpy(r + a + 1, name, b + 1);
   return r;
}
static char *dir_of(const char *path)
{
   const char *s = strrchr(path, '/');
   if (!s) return strdup("");
   char *r = malloc(s - path + 1);
   memcpy(r, path, s - path); r[s - path] = 0;
   return r;
}
static char *get_string(const OBJECT *o, const char *key, const char *dflt)
{
   char *s = NULL;
   object_get(o, key, &s, STRING, 1, dflt);
   if (!s) s = strdup("");      /* "key = ;": present and empty -- callers compare the result, none of them expects NULL */
   return s;
}
/* CRC-32 (IEEE 802.3, reflected 0xEDB88320, initial value and final xor 0xffffffff): the algorithm of the reference's
 * checksum_crc32 / checksum_crc32_table (crc32.c:46-84; its own check string "123456789" gives 0xcbf43926) */
uint32_t ddcmi_crc32(const unsigned char *p, size_t n)
{
   static uint32_t table[256];
   static int have = 0;
   if (!have)
   {
      for (uint32_t i = 0; i < 256; i++)
      {
         uint32_t c = i;
         for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
         table[i] = c;
      }
      have = 1;
   }
   uint32_t c = 0xFFFFFFFFu;
   for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xff] ^ (c >> 8);
   return c ^ 0xFFFFFFFFu;
}

static int find_name(char **names, int n, const char *name)
{
   for (int i = 0; i < n; i++) if (strcmp(names[i], name) == 0) return i;
   return -1;
}

/* CGLennardJones_setShift, bioMartini.c:840-848 */
static double lj_shift(double sigma, double eps, double rcut)
{
   double sigma_r = sigma / rcut;
   double s2 = sigma_r * sigma_r;
   double s4 = s2 * s2;
   double s6 = s4 * s2;
   double s12 = s6 * s6;
   return (-4.0 * eps * (s12 - s6));
}

/* the MMFF object tree (bioMMFF.c:9-264) */
typedef struct { int atomI, atomJ, func, valid; char *typeI, *typeJ; double kb, b0; int grp; } bondparms;   /* grp: CONSLISTPARMS index of a constraint */
typedef struct { int atomI, atomJ, atomK, func; double ktheta, theta0; } angleparms;
typedef struct { int atomI, atomJ, atomK, atomL, func, n; double kchi, delta; } torsparms;
typedef struct
{
   char *name, *resName;
   int resID, resType, centerAtom, natoms;
   char **atomName; int *atomTypeID; int *atmgrp;     /* flat over groups, atomList order */
   int nbonds; bondparms *bonds;
   int nexcl; bondparms *excl;
   int ncons; bondparms *cons;
   int nangles; angleparms *angles;
   int ntors; torsparms *tors;
} resiparms;

static int load_resi(const char *name, resiparms *r, char *err, int errlen)
{
   memset(r, 0, sizeof(*r));
   OBJECT *o = object_find(name, "RESIPARMS");
   if (!o) { snprintf(err, errlen, "RESIPARMS %s not found", name); return -1; }
   r->name = strdup(name);
   object_get(o, "resID", &r->resID, INT, 1, "0");
   object_get(o, "resType", &r->resType, INT, 1, "0");
   object_get(o, "centerAtom", &r->centerAtom, INT, 1, "0");
   r->resName = get_string(o, "resName", "NoName");
   char **groupNames = NULL;
   int ngroups = object_getv(o, "groupList", (void **)&groupNames, STRING, IGNORE_IF_NOT_FOUND);
   if (ngroups <= 0) { snprintf(err, errlen, "RESIPARMS %s: groupList missing", name); return -1; }
   int cap = 0;
   for (int g = 0; g < ngroups; g++)
   {
      OBJECT *go = object_find(groupNames[g], "GROUPPARMS");
      if (!go) { snprintf(err, errlen, "GROUPPARMS %s not found", groupNames[g]); return -1; }
      int groupID = 0;
      object_get(go, "groupID", &groupID, INT, 1, "0");
      char **atomNames = NULL;
      int na = object_getv(go, "atomList", (void **)&atomNames, STRING, IGNORE_IF_NOT_FOUND);
      cap += na;
      r->atomName = realloc(r->atomName, sizeof(char *) * cap);
      r->atomTypeID = realloc(r->atomTypeID, sizeof(int) * cap);
      r->atmgrp = realloc(r->atmgrp, sizeof(int) * cap);
      for (int a = 0; a < na; a++)
      {
         OBJECT *ao = object_find(atomNames[a], "ATOMPARMS");
         if (!ao) { snprintf(err, errlen, "ATOMPARMS %s not found", atomNames[a]); return -1; }
         int atomID = 0;
         object_get(ao, "atomID", &atomID, INT, 1, "0");
         r->atomName[r->natoms] = get_string(ao, "atomName", "NoName");
         object_get(ao, "atomTypeID", &r->atomTypeID[r->natoms], INT, 1, "0");
         r->atmgrp[r->natoms] = (groupID << 8) | (atomID & 255);   /* bioGid.h layout */
         r->natoms++;
         free(atomNames[a]);
      }
      free(atomNames);
      free(groupNames[g]);
   }
   free(groupNames);
   /* bondList / exclusionList / constraintList (bioMMFF.c:53-113,175-193) */
   for (int pass = 0; pass < 2; pass++)
   {
      char **names = NULL;
      const char *key = pass == 0 ? "bondList" : "exclusionList";
      const char *cls = pass == 0 ? "BONDPARMS" : "EXCLUDEPARMS";
      int n = object_getv(o, key, (void **)&names, STRING, IGNORE_IF_NOT_FOUND);
      bondparms *list = calloc(n > 0 ? n : 1, sizeof(bondparms));
      for (int i = 0; i < n; i++)
      {
         OBJECT *bo = object_find(names[i], cls);
         if (!bo) { snprintf(err, errlen, "%s %s not found", cls, names[i]); return -1; }
         object_get(bo, "atomI", &list[i].atomI, INT, 1, "0");
         object_get(bo, "atomJ", &list[i].atomJ, INT, 1, "0");
         object_get(bo, "func", &list[i].func, INT, 1, "1");
         list[i].typeI = get_string(bo, "atomTypeI", "NoType");
         list[i].typeJ = get_string(bo, "atomTypeJ", "NoType");
         object_get(bo, "kb", &list[i].kb, WITH_UNITS, 1, "0.0", "kJ*mol^-1*nm^-2", NULL);
         object_get(bo, "b0", &list[i].b0, WITH_UNITS, 1, "0.0", "nm", NULL);
         list[i].valid = 1;
         free(names[i]);
      }
      free(names);
      if (pass == 0) { r->nbonds = n; r->bonds = list; } else { r->nexcl = n; r->excl = list; }
   }
   {
      char **clnames = NULL;
      int ncl = object_getv(o, "constraintList", (void **)&clnames, STRING, IGNORE_IF_NOT_FOUND);
      for (int c = 0; c < ncl; c++)
      {
         OBJECT *co = object_find(clnames[c], "CONSLISTPARMS");
         if (!co) { snprintf(err, errlen, "CONSLISTPARMS %s not found", clnames[c]); return -1; }
         char **cn = NULL;
         int nc = object_getv(co, "constraintSubList", (void **)&cn, STRING, IGNORE_IF_NOT_FOUND);
         r->cons = realloc(r->cons, sizeof(bondparms) * (r->ncons + nc + 1));
         for (int i = 0; i < nc; i++)
         {
            OBJECT *po = object_find(cn[i], "CONSPARMS");
            if (!po) { snprintf(err, errlen, "CONSPARMS %s not found", cn[i]); return -1; }
            bondparms *b = &r->cons[r->ncons++];
            memset(b, 0, sizeof(*b));
            object_get(po, "atomI", &b->atomI, INT, 1, "0");
            object_get(po, "atomJ", &b->atomJ, INT, 1, "0");
            object_get(po, "func", &b->func, INT, 1, "1");
            object_get(po, "r0", &b->b0, WITH_UNITS, 1, "0.0", "nm", NULL);     /* consparms_init, bioMMFF.c:64-85 */
            b->valid = (b->func == 1);
            b->grp = c;
            free(cn[i]);
         }
         free(cn);
         free(clnames[c]);
      }
      free(clnames);
   }
   {
      char **names = NULL;
      int n = object_getv(o, "angleList", (void **)&names, STRING, IGNORE_IF_NOT_FOUND);
      r->nangles = n; r->angles = calloc(n > 0 ? n : 1, sizeof(angleparms));
      for (int i = 0; i < n; i++)
      {
         OBJECT *ao = object_find(names[i], "ANGLEPARMS");
         if (!ao) { snprintf(err, errlen, "ANGLEPARMS %s not found", names[i]); return -1; }
         object_get(ao, "atomI", &r->angles[i].atomI, INT, 1, "0");
         object_get(ao, "atomJ", &r->angles[i].atomJ, INT, 1, "0");
         object_get(ao, "atomK", &r->angles[i].atomK, INT, 1, "0");
         object_get(ao, "ktheta", &r->angles[i].ktheta, WITH_UNITS, 1, "0.0", "kJ*mol^-1", NULL);
         object_get(ao, "theta0", &r->angles[i].theta0, DOUBLE, 1, "0");
         object_get(ao, "func", &r->angles[i].func, INT, 1, "1");
         free(names[i]);
      }
      free(names);
   }
   {
      char **names = NULL;
      int n = object_getv(o, "dihedralList", (void **)&names, STRING, IGNORE_IF_NOT_FOUND);
      r->ntors = n; r->tors = calloc(n > 0 ? n : 1, sizeof(torsparms));
      for (int i = 0; i < n; i++)
      {
         OBJECT *to = object_find(names[i], "TORSPARMS");
         if (!to) { snprintf(err, errlen, "TORSPARMS %s not found", names[i]); return -1; }
         object_get(to, "atomI", &r->tors[i].atomI, INT, 1, "0");
         object_get(to, "atomJ", &r->tors[i].atomJ, INT, 1, "0");
         object_get(to, "atomK", &r->tors[i].atomK, INT, 1, "0");
         object_get(to, "atomL", &r->tors[i].atomL, INT, 1, "0");
         object_get(to, "func", &r->tors[i].func, INT, 1, "1");
         object_get(to, "n", &r->tors[i].n, INT, 1, "1");
         object_get(to, "kchi", &r->tors[i].kchi, WITH_UNITS, 1, "0.0", "kJ*mol^-1", NULL);
         object_get(to, "delta", &r->tors[i].delta, DOUBLE, 1, "0");
         free(names[i]);
      }
      free(names);
   }
   return 0;
}

static void free_resi(resiparms *r)
{
   free(r->name); free(r->resName);
   for (int i = 0; i < r->natoms; i++) free(r->atomName[i]);
   free(r->atomName); free(r->atomTypeID); free(r->atmgrp);
   for (int i = 0; i < r->nbonds; i++) { free(r->bonds[i].typeI); free(r->bonds[i].typeJ); }
   for (int i = 0; i < r->nexcl; i++) { free(r->excl[i].typeI); free(r->excl[i].typeJ); }
   free(r->bonds); free(r->excl); free(r->cons); free(r->angles); free(r->tors);
}

/* validateExclusions, bioMartini.c:54-133 */
static void validate_exclusions(resiparms *r)
{
   for (int e = 0; e < r->nexcl; e++)
      for (int i = 0; i < r->nbonds; i++)
         if (r->bonds[i].func == 1)
         {
            if ((r->excl[e].atomI == r->bonds[i].atomI && r->excl[e].atomJ == r->bonds[i].atomJ) ||
                (r->excl[e].atomI == r->bonds[i].atomJ && r->excl[e].atomJ == r->bonds[i].atomI)) { r->excl[e].valid = 0; break; }
         }
   for (int c = 0; c < r->ncons; c++)
   {
      if (r->cons[c].valid != 1) continue;
      for (int i = 0; i < r->nbonds; i++)
         if (r->bonds[i].func == 1)
         {
            if ((r->cons[c].atomI == r->bonds[i].atomI && r->cons[c].atomJ == r->bonds[i].atomJ) ||
                (r->cons[c].atomI == r->bonds[i].atomJ && r->cons[c].atomJ == r->bonds[i].atomI)) { r->cons[c].valid = 0; break; }
         }
      for (int e = 0; e < r->nexcl; e++)
         if (r->excl[e].valid == 1)
         {
            if ((r->cons[c].atomI == r->excl[e].atomI && r->cons[c].atomJ == r->excl[e].atomJ) ||
                (r->cons[c].atomI == r->excl[e].atomJ && r->cons[c].atomJ == r->excl[e].atomI)) { r->cons[c].valid = 0; break; }
         }
   }
}

/* primes.c:35-63: the next odd prime of this task's blocks.  (The reference tests primality with Montgomery products of
 * operands that are not in Montgomery form -- a strong-probable-prime test to odd bases; a prime passes it whatever the
 * bases, so below 2^64 its accepted numbers are the primes unless a composite is a strong pseudoprime to all seven.  Here:
 * Miller-Rabin to the bases 2..17, exact below 3.4e14; tests/test_oracle.py compares the two sequences.) */
static uint64_t mulmod64(uint64_t a, uint64_t b, uint64_t m) { return (uint64_t)((unsigned __int128)a * b % m); }
static int is_prime64(uint64_t n)
{
   static const uint64_t base[7] = {2, 3, 5, 7, 11, 13, 17};
   if (n < 2) return 0;
   for (int i = 0; i < 7; i++) { if (n == base[i]) return 1; if (n % base[i] == 0) return 0; }
   uint64_t d = n - 1; int r = 0;
   while ((d & 1) == 0) { d >>= 1; r++; }
   for (int i = 0; i < 7; i++)
   {
      uint64_t x = 1, b = base[i] % n, e = d;
      while (e) { if (e & 1) x = mulmod64(x, b, n); b = mulmod64(b, b, n); e >>= 1; }
      if (x == 1 || x == n - 1) continue;
      int ok = 0;
      for (int j = 1; j < r && !ok; j++) { x = mulmod64(x, x, n); ok = x == n - 1; }
      if (!ok) return 0;
   }
   return 1;
}
void ddcmi_lcg64_default(int n, const uint64_t *label, unsigned task, unsigned ntasks, uint64_t *state, uint32_t *multID, uint32_t *prime)
{
   const uint64_t blockSize = 30000, smallest = (2ull << 30) + 1ull;
   uint64_t cand = 1, upper = 0, iblock = 0, cur = 0;
   for (int i = 0; i < n; i++)
   {
      if (i % 3 == 0)      /* multID cycles 0,1,2; a new prime with every multID 0 */
      {
         do
         {
            cand += 2;
            if (cand >= upper)
            {
               upper = (iblock * ntasks + task) * blockSize + smallest;
               cand = upper - blockSize;
               if (upper % 2 == 0) upper -= 1;
               if (cand % 2 == 0) cand += 1;
               iblock++;
            }
         } while (!is_prime64(cand));
         cur = cand;
      }
      state[i] = 0x2bc6ffff8cfe166dull ^ label[i];      /* INIT_SEED ^ label */
      multID[i] = (uint32_t)(i % 3);
      prime[i] = (uint32_t)cur;
   }
}

/* atoms reader: pio FILEHEADER object + VARRECORDASCII records
 * (collection_read.c:86-200; header fields per SURVEY A.5) */
static int read_atoms(ddcmi_setup *s, const char *basepath, int nfiles_hint, char *err, int errlen)
{
   double length_convert = units_convert(1.0, "l", NULL);
   double time_convert = units_convert(1.0, "t", NULL);
   double velocity_convert = length_convert / time_convert;
   int cap = 0, n = 0;
   int nfiles = nfiles_hint > 0 ? nfiles_hint : 1;
   /* column of each quantity in a record, from the header's field_names; default = the
    * 10-field layout of the shipped decks.  Restart files written by collection_writeBLOCK
    * (collection_write.c:57-186) lead with a checksum column and may append per-particle
    * random/group state after vz, which this path does not use. */
   int col_id = 0, col_type = 2, col_group = 3, col_r = 4, col_v = 7;
   /* record checksums (collection_read.c:274-286 verifies them on read): header checksum=CRC32 + a leading
    * "checksum" column of 8 hex digits over the rest of the fixed-length record; nrecord = records in all files */
   int col_crc = -1, crc_on = 0, lrec = 0;
   long nrecord_total = -1;
   /* RANDOM type=LCG64: "state multID prime" (%llx %u %x, lcg64_parse lcg64.c:89-95) after vz; one record without it and every
    * atom gets the default values (collection_read.c:165-166,194, collection.c:100-109) */
   int rnd_field = s->random_lcg64, rnd_default = s->random_lcg64 ? 0 : 1;
   for (int f = 0; f < nfiles; f++)
   {
      char fname[4096];
      snprintf(fname, sizeof(fname), "%s%06d", basepath, f);
      FILE *fp = fopen(fname, "rb");
      if (!fp) { snprintf(err, errlen, "cannot open atoms file %s", fname); return -1; }
      fseek(fp, 0, SEEK_END); long len = ftell(fp); fseek(fp, 0, SEEK_SET);
      char *buf = malloc(len + 1);
      if (!buf) { fclose(fp); snprintf(err, errlen, "%s: out of memory reading %ld bytes", fname, len); return -1; }
      if (fread(buf, 1, len, fp) != (size_t)len) { fclose(fp); free(buf); snprintf(err, errlen, "short read on %s", fname); return -1; }
      buf[len] = 0;
      fclose(fp);
      char *p = buf;
      int nrecord_here = -1;
      if (f == 0)
      {
         long off = 0;
         OBJECT *h = object_parse_header(buf, &off);
         if (!h) { free(buf); snprintf(err, errlen, "%s: no FILEHEADER", fname); return -1; }
         char *datatype = get_string(h, "datatype", "VARRECORDASCII");
         if (strcmp(datatype, "VARRECORDASCII") != 0 && strcmp(datatype, "FIXRECORDASCII") != 0)
         {
            snprintf(err, errlen, "%s: datatype %s not supported (ASCII only)", fname, datatype);
            free(datatype); object_free(h); free(buf); return -1;
         }
         free(datatype);
         int nf = 1;
         object_get(h, "nfiles", &nf, INT, 1, "1");
         nfiles = nf;
         object_get(h, "nrecord", &nrecord_here, INT, 1, "-1");
         nrecord_total = nrecord_here;
         object_get(h, "lrec", &lrec, INT, 1, "0");
         {
            /* collection_read.c:102-108: random = NONE -> no field in the records; absent -> try to parse one */
            char *rk = get_string(h, "random", "NotSet");
            if (strcmp(rk, "NONE") == 0) rnd_field = 0;
            free(rk);
         }
         {
            char *ck = get_string(h, "checksum", "NONE");
            crc_on = strcmp(ck, "CRC32") == 0;
            free(ck);
         }
         if (object_testforkeyword(h, "field_names"))
         {
            char **names = NULL;
            int nn = object_getv(h, "field_names", (void *)&names, STRING, IGNORE_IF_NOT_FOUND);
            for (int k = 0; k < nn; k++)
            {
               if (strcmp(names[k], "checksum") == 0) col_crc = k;
               if (strcmp(names[k], "id") == 0 || strcmp(names[k], "label") == 0) col_id = k;
               else if (strcmp(names[k], "type") == 0) col_type = k;
               else if (strcmp(names[k], "group") == 0) col_group = k;
               else if (strcmp(names[k], "rx") == 0) col_r = k;
               else if (strcmp(names[k], "vx") == 0) col_v = k;
               free(names[k]);
            }
            free(names);
         }
         object_free(h);
         p = buf + off;
      }
      while (*p)
      {
         char *eol = strchr(p, '\n');
         if (eol) *eol = 0;
         char *line = p;
         while (*line && isspace((unsigned char)*line)) line++;
         if (*line)
         {
            if (crc_on && col_crc == 0 && lrec > 8 && eol && (long)(eol - p) + 1 == lrec)
            {
               /* the record as written: 8 hex digits, then lrec - 8 bytes up to and including the newline */
               *eol = '\n';
               unsigned long want = strtoul((char[]){p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], 0}, NULL, 16);
               unsigned long have = ddcmi_crc32((const unsigned char *)p + 8, (size_t)lrec - 8);
               *eol = 0;
               if (want != have) { snprintf(err, errlen, "%s: record %d fails its CRC32 (file says %08lx, bytes give %08lx): truncated or corrupted snapshot", fname, n, want, have); free(buf); return -1; }
            }
            if (n == cap)
            {
               cap = cap ? 2 * cap : 8192;
               s->rx = realloc(s->rx, sizeof(double) * cap); s->ry = realloc(s->ry, sizeof(double) * cap); s->rz = realloc(s->rz, sizeof(double) * cap);
               s->vx = realloc(s->vx, sizeof(double) * cap); s->vy = realloc(s->vy, sizeof(double) * cap); s->vz = realloc(s->vz, sizeof(double) * cap);
               s->gid = realloc(s->gid, sizeof(uint64_t) * cap);
               s->species = realloc(s->species, sizeof(int) * cap); s->group = realloc(s->group, sizeof(int) * cap);
               if (s->random_lcg64)
               {
                  s->lcg_state = realloc(s->lcg_state, sizeof(uint64_t) * cap);
                  s->lcg_multID = realloc(s->lcg_multID, sizeof(uint32_t) * cap); s->lcg_prime = realloc(s->lcg_prime, sizeof(uint32_t) * cap);
               }
            }
            /* split the record into whitespace-separated tokens (in place) */
            char *tok[32];
            int ntok = 0;
            for (char *q = line; *q && ntok < 32;)
            {
               while (*q && isspace((unsigned char)*q)) q++;
               if (!*q) break;
               tok[ntok++] = q;
               while (*q && !isspace((unsigned char)*q)) q++;
               if (*q) *q++ = 0;
            }
            int need = col_v + 3;
            if (col_r + 3 > need) need = col_r + 3;
            if (ntok < need || col_id >= ntok || col_type >= ntok || col_group >= ntok)
            { snprintf(err, errlen, "%s: record %d has %d fields, expected at least %d", fname, n, ntok, need); free(buf); return -1; }
            uint64_t label = strtoull(tok[col_id], NULL, 10);
            const char *spname = tok[col_type], *grname = tok[col_group];
            int sp = find_name(s->species_name, s->nspecies, spname);
            int gr = find_name(s->group_name, s->ngroup, grname);
            if (sp < 0 || gr < 0) { snprintf(err, errlen, "%s: unknown species '%s' or group '%s'", fname, spname, grname); free(buf); return -1; }
            double v[6];
            for (int k = 0; k < 3; k++) { v[k] = strtod(tok[col_r + k], NULL); v[3 + k] = strtod(tok[col_v + k], NULL); }
            s->gid[n] = label; s->species[n] = sp; s->group[n] = gr;
            s->rx[n] = length_convert * v[0]; s->ry[n] = length_convert * v[1]; s->rz[n] = length_convert * v[2];
            s->vx[n] = velocity_convert * v[3]; s->vy[n] = velocity_convert * v[4]; s->vz[n] = velocity_convert * v[5];
            if (s->random_lcg64)
            {
               int got = 0;
               const int c0 = col_v + 3;
               if (rnd_field && c0 + 2 < ntok)
               {
                  char *e0, *e1, *e2;
                  unsigned long long st = strtoull(tok[c0], &e0, 16);
                  unsigned long mu = strtoul(tok[c0 + 1], &e1, 10), pr = strtoul(tok[c0 + 2], &e2, 16);
                  if (!*e0 && !*e1 && !*e2 && e0 != tok[c0] && e1 != tok[c0 + 1] && e2 != tok[c0 + 2] && pr != 0 && mu <= 2)
                  { s->lcg_state[n] = st; s->lcg_multID[n] = (uint32_t)mu; s->lcg_prime[n] = (uint32_t)pr; got = 1; }
               }
               if (!got) rnd_default = 1;
            }
            n++;
         }
         if (!eol) break;
         p = eol + 1;
      }
      free(buf);
   }
   if (nrecord_total >= 0 && n != nrecord_total)
   { snprintf(err, errlen, "%s: header announces nrecord=%ld but %d records were read: truncated snapshot?", basepath, nrecord_total, n); return -1; }
   s->natoms = n;
   if (s->random_lcg64 && n > 0)
   {
      s->lcg_from_file = !rnd_default;
      if (rnd_default) ddcmi_lcg64_default(n, s->gid, 0, 1, s->lcg_state, s->lcg_multID, s->lcg_prime);
   }
   return 0;
}

ddcmi_setup *ddcmi_deck_load(const char *object_file, const char *restart_file, char *err, int errlen)
{
   return ddcmi_deck_load_with(object_file, restart_file, NULL, err, errlen);
}

ddcmi_setup *ddcmi_deck_load_with(const char *object_file, const char *restart_file, const char *extra, char *err, int errlen)
{
   ddcmi_setup *s = calloc(1, sizeof(ddcmi_setup));
   resiparms *resi = NULL;
   int nresi = 0;
   char *dir = dir_of(object_file);
   char **ljNames = NULL; int nljparms = 0;
   char **typeNames = NULL;
   char **resiNames = NULL;
   if (errlen > 0) err[0] = 0;
   units_ddcmd_defaults();
   units_clear_error();
   object_reset();
   if (object_compilefile(object_file) < 0) FAIL("cannot read object file %s", object_file);
   {
      /* objectSetup.c:40-45: object.data then the restart file */
      char *rf = restart_file ? strdup(restart_file) : path_join(dir, "restart");
      int rc = object_compilefile(rf);
      if (rc < 0 && restart_file) { free(rf); FAIL("cannot read restart file %s", restart_file); }
      free(rf);
      if (rc < 0)
      {
         /* run_ddcMD_CPU.sh links restart -> snapshot.mem/restart before the run */
         rf = path_join(dir, "snapshot.mem/restart");
         object_compilefile(rf);
         free(rf);
      }
   }
   if (extra) object_compilestring(extra);

   OBJECT *sim = object_find("simulate", "SIMULATE");
   if (!sim) FAIL("no 'simulate SIMULATE' object");
   {
      uint64_t u;
      object_get(sim, "loop", &u, U64, 1, "0"); s->loop = (int64_t)u;
      object_get(sim, "maxloop", &u, U64, 1, "0"); s->maxloop = (int64_t)u;
      int dl; object_get(sim, "deltaloop", &dl, INT, 1, "-1"); s->deltaloop = dl;
      if (dl > -1 && s->loop + dl < s->maxloop) s->maxloop = s->loop + dl;        /* simulate.c:242 */
      object_get(sim, "printrate", &s->printrate, INT, 1, "5");
      object_get(sim, "snapshotrate", &s->snapshotrate, INT, 1, "100");
      object_get(sim, "checkpointrate", &s->checkpointrate, INT, 1, "1000");
      object_get(sim, "time", &s->time, WITH_UNITS, 1, "0.0", "t", NULL);
      object_get(sim, "dt", &s->dt, WITH_UNITS, 1, "1.0", "t", NULL);
   }
   char *sysname = get_string(sim, "system", NULL);
   if (!sysname) FAIL("SIMULATE has no system key");
   OBJECT *sys = object_find(sysname, "SYSTEM");
   if (!sys) FAIL("SYSTEM %s not found", sysname);
   free(sysname);
   object_get(sys, "nConstraints", &s->nConstraints, INT, 1, "0");

   /* INTEGRATOR, ACCELERATOR */
   {
      char *iname = get_string(sim, "integrator", NULL);
      if (!iname) FAIL("SIMULATE has no integrator key");
      OBJECT *io = object_find(iname, "INTEGRATOR");
      if (!io) FAIL("INTEGRATOR %s not found", iname);
      s->integrator_type = get_string(io, "type", "NGLF");
      s->npt_isotropic = strcmp(s->integrator_type, "NGLFGPULANGEVIN") == 0;      /* changeVolumeGPUisotropic, molecularPressureGPU.cu:204-239 */
      if (strcmp(s->integrator_type, "NGLFCONSTRAINT") == 0 || s->npt_isotropic)
      {
         object_get(io, "T", &s->npt_T, WITH_UNITS, 1, "310", "T", NULL);
         object_get(io, "P0", &s->npt_P0, WITH_UNITS, 1, "0.0", "pressure", NULL);
         object_get(io, "beta", &s->npt_beta, WITH_UNITS, 1, "0.0", "1/pressure", NULL);
         object_get(io, "tauBarostat", &s->npt_tau, WITH_UNITS, 1, "0.0", "t", NULL);
      }
      free(iname);
      char *aname = get_string(sim, "accelerator", "NoAccelerator");
      OBJECT *ao = object_find(aname, "ACCELERATOR");
      s->has_accelerator = (ao != NULL);
      s->accelerator_type = ao ? get_string(ao, "type", "CUDA") : strdup("NONE");
      free(aname);
   }
   /* PRINTINFO */
   {
      char *pname = get_string(sim, "printinfo", "printinfo");
      OBJECT *po = object_find(pname, "PRINTINFO");
      s->u_pressure = get_string(po, "PRESSURE", "GPa");
      object_get(po, "printMolecularPressure", &s->printMolecularPressure, INT, 1, "0");
      object_get(po, "printStress", &s->printStress, INT, 1, "0");
      object_get(po, "printHmatrix", &s->printHmatrix, INT, 1, "0");
      s->u_energyflux = get_string(po, "ENERGYFLUX", "ueV/Ang^2/fs");
      s->u_volume = get_string(po, "VOLUME", "Ang^3");
      s->u_temperature = get_string(po, "TEMPERATURE", "K");
      s->u_energy = get_string(po, "ENERGY", "eV");
      s->u_time = get_string(po, "TIME", "fs");
      s->u_length = get_string(po, "LENGTH", "Ang");
      free(pname);
   }
   /* BOX */
   {
      char *bname = get_string(sys, "box", NULL);
      OBJECT *bo = bname ? object_find(bname, "BOX") : NULL;
      if (!bo) FAIL("BOX object not found");
      if (object_testforkeyword(bo, "bndcdn")) object_get(bo, "bndcdn", &s->pbc, INT, 1, "7");
      else object_get(bo, "pbc", &s->pbc, INT, 1, "7");
      object_get(bo, "h", s->h, WITH_UNITS, 9, "1 0 0 0 1 0 0 0 1", "l", NULL);
      free(bname);
   }
   /* NEIGHBOR, DDC */
   {
      char *nname = get_string(sys, "neighbor", NULL);
      OBJECT *no = nname ? object_find(nname, "NEIGHBOR") : NULL;
      if (!no) FAIL("NEIGHBOR object not found");
      object_get(no, "deltaR", &s->deltaR, WITH_UNITS, 1, "0", "l", NULL);
      free(nname);
      char *dname = get_string(sim, "ddc", "ddc");
      OBJECT *dobj = object_find(dname, "DDC");
      object_get(dobj, "updateRate", &s->updateRate, INT, 1, "0");
      object_get(dobj, "lx", &s->lx, INT, 1, "0");
      object_get(dobj, "ly", &s->ly, INT, 1, "0");
      object_get(dobj, "lz", &s->lz, INT, 1, "0");
      free(dname);
   }
   /* GROUPs */
   {
      char **gnames = NULL;
      s->ngroup = object_getv(sys, "groups", (void **)&gnames, STRING, IGNORE_IF_NOT_FOUND);
      if (s->ngroup <= 0) FAIL("SYSTEM has no groups");
      s->group_name = gnames;
      s->group_type = calloc(s->ngroup, sizeof(int));
      s->group_Teq = calloc(s->ngroup, sizeof(double));
      s->group_tau = calloc(s->ngroup, sizeof(double));
      s->group_vcm = calloc(3 * (size_t)s->ngroup, sizeof(double));
      s->group_interval = calloc(s->ngroup, sizeof(int));
      for (int g = 0; g < s->ngroup; g++)
      {
         OBJECT *go = object_find(gnames[g], "GROUP");
         if (!go) FAIL("GROUP %s not found", gnames[g]);
         char *type = get_string(go, "type", "");
         s->group_interval[g] = 1;
         if (strcmp(type, "FREE") == 0) s->group_type[g] = DDCMI_GROUP_FREE;
         else if (strcmp(type, "BERENDSEN") == 0)
         {
            /* berendsen_parms, berendsen.c:91-113 */
            s->group_type[g] = DDCMI_GROUP_BERENDSEN;
            object_get(go, "Teq", &s->group_Teq[g], WITH_UNITS, 1, "0.0", "T", NULL);
            char *tmp = get_string(go, "tau", "foo");
            if (strcmp(tmp, "dt") == 0) s->group_tau[g] = 0.0;
            else object_get(go, "tau", &s->group_tau[g], WITH_UNITS, 1, "1.0", "t", NULL);
            free(tmp);
            object_get(go, "interval", &s->group_interval[g], INT, 1, "1");
         }
         else if (strcmp(type, "LANGEVIN") == 0)
         {
            s->group_type[g] = DDCMI_GROUP_LANGEVIN;
            /* normalParse (langevin.c:66-91): Teq is an EQUATION of time there (eq_parse, simutil -- not in the reference tree, its
             * grammar unknown here): a constant with a unit is what this loader takes, anything else is refused, not misread */
            {
               char *teq = NULL;
               object_get(go, "Teq", &teq, LITERAL, 1, "0.0");      /* (the whole text of the value, not its first token: `300 t` is two) */
               char *endp = NULL;
               (void)strtod(teq, &endp);
               while (endp && (*endp == ' ' || *endp == '\t')) endp++;
               /* <number>[ <unit of temperature>]: whatever follows the number must convert as a temperature unit by itself
                * (a character-class check let `300-2*t` and `300*t` through as "a number with a unit") */
               int ok = endp && endp != teq;
               if (ok && *endp)
               {
                  size_t L = strlen(endp);
                  while (L > 0 && (endp[L - 1] == ' ' || endp[L - 1] == '\t' || endp[L - 1] == ';')) endp[--L] = 0;
                  if (L > 0)
                  {
                     const double one = units_convert(1.0, endp, "T");
                     ok = (one == one) && one > 0.0 && (*endp != '-' && *endp != '+' && *endp != '*' && *endp != '/' && *endp != '^');
                  }
               }
               if (!ok) { free(teq); free(type); FAIL("GROUP %s: Teq is not a constant temperature (an equation of time: set the value step by step with ddcmi_set_group_temperature)", gnames[g]); }
               free(teq);
            }
            object_get(go, "Teq", &s->group_Teq[g], WITH_UNITS, 1, "0.0", "T", NULL);
            object_get(go, "tau", &s->group_tau[g], WITH_UNITS, 1, "1.0", "t", NULL);
            object_get(go, "vcm", &s->group_vcm[3 * g], WITH_UNITS, 3, "0.0 0.0 0.0", "l/t", NULL);      /* langevin.c:167 */
            {
               /* langevin.c:71-79: GLOBAL_ENERGY steers Teq from the system's energy every step -- not a constant temperature either */
               char *dyn = get_string(go, "Teq_dynamics", "EXPLICIT_TIME");
               const int explicit_time = strcmp(dyn, "EXPLICIT_TIME") == 0;
               free(dyn);
               if (!explicit_time) { free(type); FAIL("GROUP %s: Teq_dynamics other than EXPLICIT_TIME is not supported (langevin.c:74-79 steers the temperature from the global energy)", gnames[g]); }
            }
         }
         else s->group_type[g] = DDCMI_GROUP_OTHER;
         free(type);
      }
   }
   /* RANDOM (system.c:135, random.c:44-71): type LCG64 = the particles carry their own streams (read_atoms); the seed feeds the
    * counter-based stream of decomposed runs */
   {
      char *rname = get_string(sys, "random", "NONE");
      OBJECT *ro = strcmp(rname, "NONE") != 0 ? object_find(rname, "RANDOM") : NULL;
      s->rng_seed = 0;
      if (ro)
      {
         object_get(ro, "seed", &s->rng_seed, U64, 1, "0");
         char *rtype = get_string(ro, "type", "NONE");
         if (strcmp(rtype, "LCG64") == 0) { s->random_lcg64 = 1; s->random_name = strdup(rname); }
         free(rtype);
      }
      free(rname);
   }
   /* species: via MOLECULECLASS (system.c:139-146, molecule.c:39-66,226-246) or "species" */
   {
      char *mcname = get_string(sys, "moleculeClass", "NONE");
      if (strcmp(mcname, "NONE") != 0)
      {
         OBJECT *mc = object_find(mcname, "MOLECULECLASS");
         if (!mc) FAIL("MOLECULECLASS %s not found", mcname);
         char **mnames = NULL;
         s->nmoltype = object_getv(mc, "molecules", (void **)&mnames, STRING, IGNORE_IF_NOT_FOUND);
         s->mol_nspecies = calloc(s->nmoltype > 0 ? s->nmoltype : 1, sizeof(int));
         int *own = calloc(s->nmoltype > 0 ? s->nmoltype : 1, sizeof(int));
         for (int m = 0; m < s->nmoltype; m++)
         {
            OBJECT *mo = object_find(mnames[m], "MOLECULE");
            if (!mo) FAIL("MOLECULE %s not found", mnames[m]);
            char **snames = NULL;
            int ns = object_getv(mo, "species", (void **)&snames, STRING, IGNORE_IF_NOT_FOUND);
            if (ns <= 0) FAIL("MOLECULE %s names no species", mnames[m]);      /* (its ownership species would be the next molecule's first) */
            char *ownership = get_string(mo, "ownershipSpecies", "$NONE$");
            s->mol_nspecies[m] = ns;
            own[m] = s->nspecies;       /* default: first species of the molecule */
            s->species_name = realloc(s->species_name, sizeof(char *) * (s->nspecies + ns + 1));
            s->moltype = realloc(s->moltype, sizeof(int) * (s->nspecies + ns + 1));
            for (int k = 0; k < ns; k++)
            {
               if (strcmp(snames[k], ownership) == 0) own[m] = s->nspecies;
               s->species_name[s->nspecies] = snames[k];
               s->moltype[s->nspecies] = m;
               s->nspecies++;
            }
            free(snames); free(ownership); free(mnames[m]);
         }
         free(mnames);
         /* keep the ownership species index per molecule type in bpair_off for now */
         s->bpair_off = calloc(s->nmoltype + 1, sizeof(int));
         for (int m = 0; m < s->nmoltype; m++) s->bpair_off[m] = own[m];
         free(own);
      }
      else
      {
         char **snames = NULL;
         s->nspecies = object_getv(sys, "species", (void **)&snames, STRING, IGNORE_IF_NOT_FOUND);
         if (s->nspecies <= 0) FAIL("SYSTEM has neither moleculeClass nor species");
         s->species_name = snames;
         s->moltype = calloc(s->nspecies, sizeof(int));
         s->nmoltype = 0;
      }
      free(mcname);
      s->mass = calloc(s->nspecies, sizeof(double));
      s->charge = calloc(s->nspecies, sizeof(double));
      s->ljtype = calloc(s->nspecies, sizeof(int));
      s->resitype = calloc(s->nspecies, sizeof(int));
      s->atomoffset = calloc(s->nspecies, sizeof(int));
      for (int i = 0; i < s->nspecies; i++)
      {
         OBJECT *so = object_find(s->species_name[i], "SPECIES");
         if (!so) FAIL("SPECIES %s not found", s->species_name[i]);
         object_get(so, "mass", &s->mass[i], WITH_UNITS, 1, "1.0", "m", NULL);       /* species.c:35 */
         object_get(so, "charge", &s->charge[i], WITH_UNITS, 1, "0.0", "i*t", NULL); /* species.c:36 */
      }
   }
   /* POTENTIAL type=MARTINI */
   {
      char **pnames = NULL;
      int np = object_getv(sys, "potential", (void **)&pnames, STRING, IGNORE_IF_NOT_FOUND);
      OBJECT *pot = NULL, *rpot = NULL;
      for (int i = 0; i < np; i++)
      {
         OBJECT *po = object_find(pnames[i], "POTENTIAL");
         if (po)
         {
            char *type = get_string(po, "type", "");
            if (strcmp(type, "MARTINI") == 0 && !pot) pot = po;
            if (strcmp(type, "RESTRAINT") == 0 && !rpot) rpot = po;
            free(type);
         }
         free(pnames[i]);
      }
      free(pnames);
      if (!pot) FAIL("no POTENTIAL of type MARTINI in SYSTEM potential list");
      char *parmfile = get_string(pot, "parmfile", "martini.data");
      char *pf = path_join(dir, parmfile);
      if (object_compilefile(pf) < 0) { free(pf); FAIL("cannot read parmfile %s", parmfile); }
      free(pf); free(parmfile);
      object_get(pot, "excludePotentialTerm", &s->excludePotentialTerm, INT, 1, "0");
      object_get(pot, "cutoff", &s->rmax, WITH_UNITS, 1, "11.0", "Angstrom", NULL);
      object_get(pot, "potential-shift", &s->potentialShift, INT, 1, "1");
      object_get(pot, "rcoulomb", &s->rcoulomb, WITH_UNITS, 1, "11.0", "Angstrom", NULL);
      object_get(pot, "epsilon_r", &s->epsilon_r, DOUBLE, 1, "15.0");
      object_get(pot, "epsilon_rf", &s->epsilon_rf, DOUBLE, 1, "-1.0");
      /* bioMartini.c:1234-1245 */
      double irc = 1.0 / s->rcoulomb;
      double irc3 = irc * irc * irc;
      if (s->epsilon_rf != -1.0)
      {
         s->krf = (s->epsilon_rf - s->epsilon_r) / (2 * s->epsilon_rf + s->epsilon_r) * irc3;
         s->crf = 3 * (s->epsilon_rf) / (2 * s->epsilon_rf + s->epsilon_r) * irc;
      }
      else
      {
         s->krf = 0.5 * irc3;
         s->crf = 1.5 * irc;
      }
      s->keR = units_ke() / s->epsilon_r;   /* bioMartini.c:1033 */
      /* POTENTIAL type=RESTRAINT (restraint.c:177-208, :57-120): parmfile -> RESTRAINTLIST "restraint" */
      if (rpot)
      {
         char *rfile = get_string(rpot, "parmfile", "restraint.data");
         char *rf = path_join(dir, rfile);
         if (object_compilefile(rf) < 0) { free(rf); FAIL("cannot read restraint parmfile %s", rfile); }
         free(rf); free(rfile);
         OBJECT *rl = object_find("restraint", "RESTRAINTLIST");
         if (!rl) FAIL("RESTRAINTLIST object 'restraint' not found");
         object_get(rl, "origin", &s->rest_origin, INT, 1, "0");
         char **rnames = NULL;
         int nr = object_getv(rl, "restraintList", (void **)&rnames, STRING, IGNORE_IF_NOT_FOUND);
         s->nrest = nr > 0 ? nr : 0;
         s->rest_gid = calloc(s->nrest + 1, sizeof(uint64_t));
         s->rest_fc = calloc(3 * s->nrest + 3, sizeof(int));
         s->rest_r0 = calloc(3 * s->nrest + 3, sizeof(double));
         s->rest_kb = calloc(s->nrest + 1, sizeof(double));
         for (int r = 0; r < s->nrest; r++)
         {
            OBJECT *ro = object_find(rnames[r], "RESTRAINTPARMS");
            if (!ro) FAIL("RESTRAINTPARMS %s not found", rnames[r]);
            object_get(ro, "gid", &s->rest_gid[r], U64, 1, "0");
            object_get(ro, "fcx", &s->rest_fc[3 * r], INT, 1, "0");
            object_get(ro, "fcy", &s->rest_fc[3 * r + 1], INT, 1, "0");
            object_get(ro, "fcz", &s->rest_fc[3 * r + 2], INT, 1, "0");
            object_get(ro, "x0", &s->rest_r0[3 * r], DOUBLE, 1, "0");
            object_get(ro, "y0", &s->rest_r0[3 * r + 1], DOUBLE, 1, "0");
            object_get(ro, "z0", &s->rest_r0[3 * r + 2], DOUBLE, 1, "0");
            object_get(ro, "kb", &s->rest_kb[r], WITH_UNITS, 1, "0.0", "kJ*mol^-1*nm^-2", NULL);
            free(rnames[r]);
         }
         free(rnames);
      }
   }
   /* MMFF (bioMMFF.c:236-270): object named "martini" */
   OBJECT *mmff = object_find("martini", "MMFF");
   if (!mmff) FAIL("no 'martini MMFF' object in the parmfile");
   nresi = object_getv(mmff, "resiParms", (void **)&resiNames, STRING, IGNORE_IF_NOT_FOUND);
   s->nlj = object_getv(mmff, "atomTypeList", (void **)&typeNames, STRING, IGNORE_IF_NOT_FOUND);
   nljparms = object_getv(mmff, "ljParms", (void **)&ljNames, STRING, IGNORE_IF_NOT_FOUND);
   if (nresi <= 0 || s->nlj <= 0) FAIL("MMFF needs resiParms and atomTypeList");
   resi = calloc(nresi, sizeof(resiparms));
   for (int r = 0; r < nresi; r++)
   {
      if (load_resi(resiNames[r], &resi[r], err, errlen) != 0) goto fail;
      validate_exclusions(&resi[r]);
   }
   /* LJ table (martiniLJ_parms bioMartini.c:868-950) */
   {
      int n2 = s->nlj * s->nlj;
      s->sigma = malloc(sizeof(double) * n2); s->eps = malloc(sizeof(double) * n2); s->shift = malloc(sizeof(double) * n2);
      for (int i = 0; i < n2; i++) { s->sigma[i] = NAN; s->eps[i] = NAN; s->shift[i] = NAN; }
      for (int i = 0; i < nljparms; i++)
      {
         OBJECT *lo = object_find(ljNames[i], "LJPARMS");
         if (!lo) FAIL("LJPARMS %s not found", ljNames[i]);
         int indexI, indexJ; double sigma, eps;
         object_get(lo, "indexI", &indexI, INT, 1, "0");
         object_get(lo, "indexJ", &indexJ, INT, 1, "0");
         object_get(lo, "sigma", &sigma, WITH_UNITS, 1, "1.0", "nm", NULL);
         object_get(lo, "eps", &eps, WITH_UNITS, 1, "0.0", "kJ*mol^-1", NULL);
         if (indexI < 0 || indexJ < 0 || indexI >= s->nlj || indexJ >= s->nlj) FAIL("LJPARMS %s: index out of range", ljNames[i]);
         int ab = indexI + indexJ * s->nlj, ba = indexJ + indexI * s->nlj;
         s->sigma[ab] = s->sigma[ba] = sigma;
         s->eps[ab] = s->eps[ba] = eps;
         s->shift[ab] = s->shift[ba] = s->potentialShift ? lj_shift(sigma, eps, s->rmax) : 0.0;
      }
   }
   /* species -> residue, atom offset, LJ type (getCGLJindexbySpecie: the residue
    * name is the species name cut at the first of 'c','n','x') */
   for (int i = 0; i < s->nspecies; i++)
   {
      const char *sp = s->species_name[i];
      const char *delim = strpbrk(sp, "cnx");
      if (!delim) FAIL("species name %s has no residue delimiter (c,n,x)", sp);
      size_t rl = delim - sp;
      const char *atmName = delim + 1;
      int found = 0;
      for (int r = 0; r < nresi && !found; r++)
      {
         if (strlen(resi[r].resName) != rl || strncmp(resi[r].resName, sp, rl) != 0) continue;
         for (int a = 0; a < resi[r].natoms; a++)
            if (strcmp(resi[r].atomName[a], atmName) == 0)
            {
               s->resitype[i] = r; s->atomoffset[i] = a; s->ljtype[i] = resi[r].atomTypeID[a];
               found = 1; break;
            }
      }
      if (!found) FAIL("getCGLJindexbySpecie: no residue/atom for species %s", sp);
   }
   /* residue bonded tables (flat) */
   {
      s->nresi = nresi;
      s->resi_natoms = calloc(nresi, sizeof(int));
      s->bond_off = calloc(nresi + 1, sizeof(int));
      s->angle_off = calloc(nresi + 1, sizeof(int));
      s->tors_off = calloc(nresi + 1, sizeof(int));
      int nb = 0, na = 0, nt = 0;
      for (int r = 0; r < nresi; r++) { nb += resi[r].nbonds; na += resi[r].nangles; nt += resi[r].ntors; }
      s->bondI = calloc(nb + 1, sizeof(int)); s->bondJ = calloc(nb + 1, sizeof(int));
      s->bond_kb = calloc(nb + 1, sizeof(double)); s->bond_b0 = calloc(nb + 1, sizeof(double));
      s->angleI = calloc(na + 1, sizeof(int)); s->angleJ = calloc(na + 1, sizeof(int)); s->angleK = calloc(na + 1, sizeof(int));
      s->angle_func = calloc(na + 1, sizeof(int)); s->angle_k = calloc(na + 1, sizeof(double)); s->angle_t0 = calloc(na + 1, sizeof(double));
      s->torsI = calloc(nt + 1, sizeof(int)); s->torsJ = calloc(nt + 1, sizeof(int)); s->torsK = calloc(nt + 1, sizeof(int)); s->torsL = calloc(nt + 1, sizeof(int));
      s->tors_func = calloc(nt + 1, sizeof(int)); s->tors_n = calloc(nt + 1, sizeof(int));
      s->tors_k = calloc(nt + 1, sizeof(double)); s->tors_delta = calloc(nt + 1, sizeof(double));
      nb = na = nt = 0;
      for (int r = 0; r < nresi; r++)
      {
         s->resi_natoms[r] = resi[r].natoms;
         s->bond_off[r] = nb; s->angle_off[r] = na; s->tors_off[r] = nt;
         /* every BONDPARMS becomes a harmonic bond term (genMartiniConn :640-656) */
         for (int i = 0; i < resi[r].nbonds; i++, nb++)
         {
            s->bondI[nb] = resi[r].bonds[i].atomI; s->bondJ[nb] = resi[r].bonds[i].atomJ;
            s->bond_kb[nb] = resi[r].bonds[i].kb; s->bond_b0[nb] = resi[r].bonds[i].b0;
         }
         for (int i = 0; i < resi[r].nangles; i++)
         {
            int f = resi[r].angles[i].func;
            if (f != 1 && f != 2 && f != 10) continue;    /* genMartiniConn :660-676 keeps only these */
            s->angleI[na] = resi[r].angles[i].atomI; s->angleJ[na] = resi[r].angles[i].atomJ; s->angleK[na] = resi[r].angles[i].atomK;
            s->angle_func[na] = f; s->angle_k[na] = resi[r].angles[i].ktheta; s->angle_t0[na] = resi[r].angles[i].theta0;
            na++;
         }
         for (int i = 0; i < resi[r].ntors; i++)
         {
            int f = resi[r].tors[i].func;
            if (f != 1 && f != 2) continue;               /* :751-763 */
            s->torsI[nt] = resi[r].tors[i].atomI; s->torsJ[nt] = resi[r].tors[i].atomJ;
            s->torsK[nt] = resi[r].tors[i].atomK; s->torsL[nt] = resi[r].tors[i].atomL;
            s->tors_func[nt] = f; s->tors_n[nt] = resi[r].tors[i].n;
            s->tors_k[nt] = resi[r].tors[i].kchi; s->tors_delta[nt] = resi[r].tors[i].delta;
            nt++;
         }
      }
      s->bond_off[nresi] = nb; s->angle_off[nresi] = na; s->tors_off[nresi] = nt;
      /* constraint groups (genConstraint, bioMartini.c:300-445): one group per CONSLISTPARMS, pairs in deck order */
      int nc = 0;
      for (int r = 0; r < nresi; r++) nc += resi[r].ncons;
      s->cons_off = calloc(nresi + 1, sizeof(int));
      s->consI = calloc(nc + 1, sizeof(int)); s->consJ = calloc(nc + 1, sizeof(int)); s->cons_grp = calloc(nc + 1, sizeof(int));
      s->cons_r0 = calloc(nc + 1, sizeof(double));
      nc = 0;
      for (int r = 0; r < nresi; r++)
      {
         s->cons_off[r] = nc;
         for (int i = 0; i < resi[r].ncons; i++, nc++)
         { s->consI[nc] = resi[r].cons[i].atomI; s->consJ[nc] = resi[r].cons[i].atomJ; s->cons_grp[nc] = resi[r].cons[i].grp; s->cons_r0[nc] = resi[r].cons[i].b0; }
      }
      s->cons_off[nresi] = nc;
   }
   /* molecule type -> bpair list of the ownership species' residue
    * (reOrgPairs bioMartini.c:1416-1423; genMartiniBondPair :135-282).  The
    * reference compares (label & atmgrpMask) with the raw atomI/atomJ of the
    * deck, so the raw values are kept. */
   if (s->nmoltype > 0)
   {
      int *own = malloc(sizeof(int) * s->nmoltype);
      for (int m = 0; m < s->nmoltype; m++) own[m] = s->bpair_off[m];
      int total = 0;
      for (int m = 0; m < s->nmoltype; m++)
      {
         if (own[m] < 0 || own[m] >= s->nspecies || s->resitype[own[m]] < 0 || s->resitype[own[m]] >= nresi)
         { free(own); FAIL("molecule type %d: its ownership species has no residue in the parameter file", m); }
         resiparms *r = &resi[s->resitype[own[m]]];
         total += r->nbonds + r->nexcl + r->ncons;
         s->nresicons += r->ncons;
      }
      s->bpairI = calloc(total + 1, sizeof(int)); s->bpairJ = calloc(total + 1, sizeof(int));
      int k = 0;
      for (int m = 0; m < s->nmoltype; m++)
      {
         resiparms *r = &resi[s->resitype[own[m]]];
         s->bpair_off[m] = k;
         for (int i = 0; i < r->nbonds; i++) if (r->bonds[i].func == 1) { s->bpairI[k] = r->bonds[i].atomI; s->bpairJ[k] = r->bonds[i].atomJ; k++; }
         for (int i = 0; i < r->nexcl; i++) if (r->excl[i].valid == 1) { s->bpairI[k] = r->excl[i].atomI; s->bpairJ[k] = r->excl[i].atomJ; k++; }
         for (int i = 0; i < r->ncons; i++) if (r->cons[i].valid == 1) { s->bpairI[k] = r->cons[i].atomI; s->bpairJ[k] = r->cons[i].atomJ; k++; }
      }
      s->bpair_off[s->nmoltype] = k;
      free(own);
   }
   /* COLLECTION (collection.c:77-86): atoms files */
   {
      char *cname = get_string(sys, "collection", NULL);
      OBJECT *co = cname ? object_find(cname, "COLLECTION") : NULL;
      if (!co) FAIL("COLLECTION object not found (is the restart file present?)");
      char *files = get_string(co, "files", "snapshot.mem/atoms#");
      /* ddcMD resolves the path against its working directory = the deck directory.  Tried in
       * turn: the object file's directory, the working directory, the restart file's directory and its parent */
      char *base = path_join(dir, files);
      char probe[4200];
      snprintf(probe, sizeof(probe), "%s%06d", base, 0);
      if (files[0] != '/' && access(probe, R_OK) != 0)
      {
         snprintf(probe, sizeof(probe), "%s%06d", files, 0);
         if (access(probe, R_OK) == 0) { free(base); base = strdup(files); }
         else if (restart_file)
         {
            char *rdir = dir_of(restart_file);
            char *alt = path_join(rdir, files);
            snprintf(probe, sizeof(probe), "%s%06d", alt, 0);
            if (access(probe, R_OK) == 0) { free(base); base = alt; }
            else
            {
               /* snapshot.<loop>/restart names snapshot.<loop>/atoms# relative to the run directory */
               free(alt);
               char *rrdir = dir_of(rdir);
               alt = path_join(rrdir, files);
               snprintf(probe, sizeof(probe), "%s%06d", alt, 0);
               if (access(probe, R_OK) == 0) { free(base); base = alt; } else free(alt);
               free(rrdir);
            }
            free(rdir);
         }
      }
      int rc = read_atoms(s, base, 0, err, errlen);
      free(base); free(files); free(cname);
      if (rc != 0) goto fail;
      int size = -1;
      object_get(co, "size", &size, INT, 1, "-1");
      if (size >= 0 && size != s->natoms) FAIL("COLLECTION size=%d but %d records read", size, s->natoms);
   }
   /* a value whose unit could not be read came back as NaN somewhere above: the deck is refused, not run with it (tools/fuzz_decks.py, round 6) */
   if (units_error()[0]) FAIL("%s", units_error());
   for (int r = 0; r < nresi; r++) { free_resi(&resi[r]); free(resiNames[r]); }
   free(resi); free(resiNames);
   for (int i = 0; i < s->nlj; i++) free(typeNames[i]);
   free(typeNames);
   for (int i = 0; i < nljparms; i++) free(ljNames[i]);
   free(ljNames);
   free(dir);
   return s;
fail:
   free(dir);
   ddcmi_setup_free(s);
   return NULL;
}

void ddcmi_setup_free(ddcmi_setup *s)
{
   if (!s) return;
   free(s->sigma); free(s->eps); free(s->shift);
   for (int i = 0; i < s->nspecies; i++) if (s->species_name) free(s->species_name[i]);
   free(s->species_name); free(s->mass); free(s->charge); free(s->ljtype); free(s->moltype); free(s->resitype); free(s->atomoffset);
   free(s->mol_nspecies); free(s->bpair_off); free(s->bpairI); free(s->bpairJ);
   free(s->resi_natoms); free(s->bond_off); free(s->bondI); free(s->bondJ); free(s->bond_kb); free(s->bond_b0);
   free(s->angle_off); free(s->angleI); free(s->angleJ); free(s->angleK); free(s->angle_func); free(s->angle_k); free(s->angle_t0);
   free(s->tors_off); free(s->torsI); free(s->torsJ); free(s->torsK); free(s->torsL); free(s->tors_func); free(s->tors_n); free(s->tors_k); free(s->tors_delta);
   for (int i = 0; i < s->ngroup; i++) if (s->group_name) free(s->group_name[i]);
   free(s->group_name); free(s->group_type); free(s->group_Teq); free(s->group_tau); free(s->group_vcm); free(s->group_interval);
   free(s->rx); free(s->ry); free(s->rz); free(s->vx); free(s->vy); free(s->vz); free(s->gid); free(s->species); free(s->group);
   free(s->rest_gid); free(s->rest_fc); free(s->rest_r0); free(s->rest_kb);
   free(s->cons_off); free(s->consI); free(s->consJ); free(s->cons_grp); free(s->cons_r0);
   free(s->integrator_type); free(s->accelerator_type);
   free(s->u_energyflux);
   free(s->random_name); free(s->lcg_state); free(s->lcg_multID); free(s->lcg_prime);
   free(s->u_pressure); free(s->u_volume); free(s->u_temperature); free(s->u_energy); free(s->u_time); free(s->u_length);
   free(s);
}
