/* object.c -- see object.h.  Clean-room reader for ddcMD object files. */
#include "object.h"
#include "units.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <stdarg.h>
#include <math.h>

static OBJECT **db = NULL;
static int ndb = 0, mdb = 0;

static char *xstrndup(const char *s, size_t n)
{
   char *r = malloc(n + 1);
   memcpy(r, s, n);
   r[n] = 0;
   return r;
}

void object_free(OBJECT *o)
{
   if (!o) return;
   free(o->name); free(o->objclass); free(o->value); free(o);
}
void object_reset(void)
{
   for (int i = 0; i < ndb; i++) object_free(db[i]);
   free(db);
   db = NULL; ndb = mdb = 0;
}

/* strip // and C-style comments in place (comment text becomes blanks) */
static char *strip_comments(const char *text)
{
   size_t n = strlen(text);
   char *out = malloc(n + 1);
   size_t i = 0;
   while (i < n)
   {
      if (text[i] == '/' && i + 1 < n && text[i + 1] == '/')
      {
         while (i < n && text[i] != '\n') out[i++] = ' ';
      }
      else if (text[i] == '/' && i + 1 < n && text[i + 1] == '*')
      {
         out[i] = ' '; out[i + 1] = ' '; i += 2;
         while (i < n && !(text[i] == '*' && i + 1 < n && text[i + 1] == '/')) { out[i] = (text[i] == '\n') ? '\n' : ' '; i++; }
         if (i < n) { out[i] = ' '; out[i + 1] = ' '; i += 2; }
      }
      else { out[i] = text[i]; i++; }
   }
   out[n] = 0;
   return out;
}

/* parse one "name CLASS { body }" at *pp; returns NULL at end of text */
static OBJECT *parse_one(const char **pp)
{
   const char *p = *pp;
   while (*p && isspace((unsigned char)*p)) p++;
   if (!*p) return NULL;
   const char *n0 = p;
   while (*p && !isspace((unsigned char)*p) && *p != '{') p++;
   const char *n1 = p;
   while (*p && isspace((unsigned char)*p)) p++;
   const char *c0 = p;
   while (*p && !isspace((unsigned char)*p) && *p != '{') p++;
   const char *c1 = p;
   while (*p && *p != '{') p++;
   if (*p != '{' || n1 == n0 || c1 == c0) { *pp = p + strlen(p); return NULL; }
   p++;
   const char *b0 = p;
   while (*p && *p != '}') p++;
   const char *b1 = p;
   if (*p == '}') p++;
   *pp = p;
   OBJECT *o = calloc(1, sizeof(OBJECT));
   o->name = xstrndup(n0, n1 - n0);
   o->objclass = xstrndup(c0, c1 - c0);
   o->value = xstrndup(b0, b1 - b0);
   return o;
}

static void db_insert(OBJECT *o)
{
   for (int i = 0; i < ndb; i++)
      if (strcmp(db[i]->name, o->name) == 0 && strcmp(db[i]->objclass, o->objclass) == 0)
      {
         /* later definition extends the earlier; later keys win (see find_key) */
         size_t a = strlen(db[i]->value), b = strlen(o->value);
         db[i]->value = realloc(db[i]->value, a + b + 3);
         /* make sure the old body ends with ';' so the keys do not run together */
         memcpy(db[i]->value + a, " ;", 2);
         memcpy(db[i]->value + a + 2, o->value, b + 1);
         object_free(o);
         return;
      }
   if (ndb == mdb) { mdb = mdb ? 2 * mdb : 64; db = realloc(db, sizeof(OBJECT *) * mdb); }
   db[ndb++] = o;
}

int object_compilestring(const char *text)
{
   char *clean = strip_comments(text);
   const char *p = clean;
   int count = 0;
   OBJECT *o;
   while ((o = parse_one(&p)) != NULL) { db_insert(o); count++; }
   free(clean);
   return count;
}

static char *read_file(const char *filename, long *len)
{
   FILE *f = fopen(filename, "rb");
   if (!f) return NULL;
   fseek(f, 0, SEEK_END);
   long n = ftell(f);
   fseek(f, 0, SEEK_SET);
   char *buf = malloc(n + 1);
   if (fread(buf, 1, n, f) != (size_t)n) { fclose(f); free(buf); return NULL; }
   buf[n] = 0;
   fclose(f);
   if (len) *len = n;
   return buf;
}

int object_compilefile(const char *filename)
{
   char *buf = read_file(filename, NULL);
   if (!buf) return -1;
   int c = object_compilestring(buf);
   free(buf);
   return c;
}

OBJECT *object_parse_header(const char *text, long *data_offset)
{
   const char *p = text;
   OBJECT *o = parse_one(&p);
   if (!o) return NULL;
   /* records start after the header's closing brace and the following blank line(s) */
   while (*p && (*p == ' ' || *p == '\t' || *p == '\r')) p++;
   while (*p == '\n' || *p == '\r') p++;
   if (data_offset) *data_offset = (long)(p - text);
   return o;
}

OBJECT *object_find(const char *name, const char *objclass)
{
   for (int i = 0; i < ndb; i++)
      if (strcmp(db[i]->name, name) == 0 && strcmp(db[i]->objclass, objclass) == 0) return db[i];
   return NULL;
}
OBJECT *object_find_byname(const char *name)
{
   for (int i = 0; i < ndb; i++)
      if (strcmp(db[i]->name, name) == 0) return db[i];
   return NULL;
}
int object_exists(const char *name, const char *objclass) { return object_find(name, objclass) != NULL; }

/* locate the LAST "key = value;" in the body; returns malloc'ed value text or NULL */
static char *find_key(const OBJECT *obj, const char *key)
{
   if (!obj || !obj->value) return NULL;
   const char *p = obj->value;
   size_t klen = strlen(key);
   char *result = NULL;
   while (*p)
   {
      /* statement = up to ';' */
      const char *s0 = p;
      while (*p && *p != ';') p++;
      const char *s1 = p;
      if (*p == ';') p++;
      const char *q = s0;
      while (q < s1 && isspace((unsigned char)*q)) q++;
      const char *k0 = q;
      while (q < s1 && !isspace((unsigned char)*q) && *q != '=') q++;
      size_t kl = q - k0;
      while (q < s1 && isspace((unsigned char)*q)) q++;
      if (q < s1 && *q == '=' && kl == klen && strncmp(k0, key, klen) == 0)
      {
         q++;
         while (q < s1 && isspace((unsigned char)*q)) q++;
         const char *e = s1;
         while (e > q && isspace((unsigned char)e[-1])) e--;
         free(result);
         result = xstrndup(q, e - q);
      }
   }
   return result;
}

int object_testforkeyword(const OBJECT *obj, const char *key)
{
   char *v = find_key(obj, key);
   if (!v) return 0;
   free(v);
   return 1;
}

/* split on whitespace; returns count, tokens malloc'ed */
static int tokenize(const char *s, char ***ptok)
{
   int n = 0, m = 8;
   char **tok = malloc(sizeof(char *) * m);
   const char *p = s;
   while (*p)
   {
      while (*p && isspace((unsigned char)*p)) p++;
      if (!*p) break;
      const char *t0 = p;
      if (*p == '"')
      {
         t0 = ++p;
         while (*p && *p != '"') p++;
         if (n == m) { m *= 2; tok = realloc(tok, sizeof(char *) * m); }
         tok[n++] = xstrndup(t0, p - t0);
         if (*p == '"') p++;
         continue;
      }
      while (*p && !isspace((unsigned char)*p)) p++;
      if (n == m) { m *= 2; tok = realloc(tok, sizeof(char *) * m); }
      tok[n++] = xstrndup(t0, p - t0);
   }
   *ptok = tok;
   return n;
}
static void free_tokens(char **tok, int n)
{
   for (int i = 0; i < n; i++) free(tok[i]);
   free(tok);
}

int object_keywordSize(const OBJECT *obj, const char *key)
{
   char *v = find_key(obj, key);
   if (!v) return 0;
   char **tok; int n = tokenize(v, &tok);
   free_tokens(tok, n); free(v);
   return n;
}

/* WITH_UNITS value list: numbers, optionally each followed by a unit expression,
 * or one trailing unit for the whole list.  Token forms: "11.0", "Angstrom",
 * "310K", "3.0e-4/bar", "72.0M_p", "kJ*mol^-1". */
static int parse_with_units(const char *text, double *out, int nmax, const char *default_unit)
{
   char **tok; int nt = tokenize(text, &tok);
   int n = 0;
   double raw[64]; char *unit[64];
   for (int i = 0; i < 64; i++) unit[i] = NULL;
   for (int i = 0; i < nt && n <= 64; i++)
   {
      char *end;
      double v = strtod(tok[i], &end);
      if (end != tok[i])
      {
         if (n == 64) break;
         raw[n] = v;
         if (*end) unit[n] = strdup(end);        /* "310K", "3.0e-4/bar" */
         n++;
      }
      else if (n > 0)
      {
         /* a bare unit token applies to the preceding number(s) without a unit */
         if (unit[n - 1] == NULL)
         {
            unit[n - 1] = strdup(tok[i]);
         }
         else
         {
            size_t a = strlen(unit[n - 1]), b = strlen(tok[i]);
            unit[n - 1] = realloc(unit[n - 1], a + b + 2);
            memcpy(unit[n - 1] + a, tok[i], b + 1);
         }
      }
   }
   /* a single trailing unit covers the whole list (e.g. "h = 1 0 0 0 1 0 0 0 1 Angstrom") */
   const char *trailing = (n > 0) ? unit[n - 1] : NULL;
   int nout = n < nmax ? n : nmax;
   for (int i = 0; i < nout; i++)
   {
      const char *u = unit[i] ? unit[i] : (trailing ? trailing : default_unit);
      if (u && u[0] == '/')
      {
         char tmp[128]; snprintf(tmp, sizeof(tmp), "1%s", u);
         out[i] = units_convert(raw[i], tmp, NULL);
      }
      else out[i] = units_convert(raw[i], u, NULL);
   }
   for (int i = 0; i < 64; i++) free(unit[i]);
   free_tokens(tok, nt);
   return nout;
}

static int store_tokens(char **tok, int nt, void *ptr, int type, int n)
{
   int cnt = nt < n ? nt : n;
   for (int i = 0; i < cnt; i++)
   {
      switch (type)
      {
      case STRING: case LITERAL: ((char **)ptr)[i] = strdup(tok[i]); break;
      case INT: ((int *)ptr)[i] = (int)strtol(tok[i], NULL, 10); break;
      case DOUBLE: ((double *)ptr)[i] = strtod(tok[i], NULL); break;
      case U64: ((uint64_t *)ptr)[i] = strtoull(tok[i], NULL, 10); break;
      default: break;
      }
   }
   return cnt;
}

int object_get(const OBJECT *obj, const char *key, void *ptr, int type, int n, const char *dflt, ...)
{
   const char *default_unit = NULL;
   if (type == WITH_UNITS)
   {
      va_list ap; va_start(ap, dflt);
      default_unit = va_arg(ap, const char *);
      va_end(ap);
   }
   char *v = find_key(obj, key);
   const char *text = v ? v : dflt;
   int cnt = 0;
   if (text)
   {
      if (type == WITH_UNITS) cnt = parse_with_units(text, (double *)ptr, n, default_unit);
      else if (type == LITERAL) { ((char **)ptr)[0] = strdup(text); cnt = 1; }
      else
      {
         char **tok; int nt = tokenize(text, &tok);
         cnt = store_tokens(tok, nt, ptr, type, n);
         free_tokens(tok, nt);
      }
   }
   free(v);
   return cnt;
}

int object_getv(const OBJECT *obj, const char *key, void **ptr, int type, int flag)
{
   char *v = find_key(obj, key);
   if (!v)
   {
      *ptr = NULL;
      if (flag == ABORT_IF_NOT_FOUND)
      {
         fprintf(stderr, "object_getv: keyword '%s' not found in object %s %s\n", key, obj ? obj->name : "(null)", obj ? obj->objclass : "");
         exit(1);
      }
      return 0;
   }
   char **tok; int nt = tokenize(v, &tok);
   int ntok = nt;
   size_t sz = (type == STRING) ? sizeof(char *) : (type == INT) ? sizeof(int) : (type == U64) ? sizeof(uint64_t) : sizeof(double);
   void *buf = malloc(sz * (nt > 0 ? nt : 1));
   if (type == WITH_UNITS) nt = parse_with_units(v, (double *)buf, nt, NULL);
   else store_tokens(tok, nt, buf, type, nt);
   free_tokens(tok, ntok);
   free(v);
   *ptr = buf;
   return nt;
}
