/*
 * object.h -- ddcMD "object database" reader (clean-room).
 *
 * The reference gets this from LLNL/simutil's object.c/.h, which is an empty
 * submodule in the checkout (src/object.c is a dangling symlink), so the
 * grammar is reconstructed from its call sites and from the shipped decks:
 *
 *    name CLASS { key = v1 v2 ...; key2 = ...; }        // comment
 *
 * - a later block with the same (name, CLASS) extends/overrides the earlier one
 *   (object.data, then the restart file: objectSetup.c:40-45),
 * - values may carry a trailing unit expression ("11.0 Angstrom", "310K",
 *   "3.0e-4/bar", "5.6 kJ*mol^-1", "72.0 M_p"),
 * - values may span lines (BOX h = 3x3 numbers).
 *
 * Call signatures follow the reference's use: object_get(obj, key, ptr, TYPE, n,
 * default [, default_unit, NULL]) and object_getv(obj, key, &ptr, TYPE, flag)
 * (e.g. bioMartini.c:1216-1233, bioMMFF.c:9-264).
 */
#ifndef DDCMI_OBJECT_H
#define DDCMI_OBJECT_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum OBJECTTYPES { STRING = 1, INT, DOUBLE, U64, WITH_UNITS, LITERAL };
enum OBJECT_GETV_FLAGS { ABORT_IF_NOT_FOUND = 1, IGNORE_IF_NOT_FOUND, EMPTY_IF_NOT_FOUND };

typedef struct object_st
{
   char *name;
   char *objclass;
   char *value;      /* concatenated "key=...;" bodies */
} OBJECT;

/* compile a file / an in-memory string into the database; returns number of
 * objects read or -1 (file missing) */
int object_compilefile(const char *filename);
int object_compilestring(const char *text);
void object_reset(void);   /* drop the whole database */

OBJECT *object_find(const char *name, const char *objclass);   /* NULL if absent */
OBJECT *object_find_byname(const char *name);                  /* first object with that name */
int object_exists(const char *name, const char *objclass);
int object_testforkeyword(const OBJECT *obj, const char *key);
int object_keywordSize(const OBJECT *obj, const char *key);    /* number of values */

/* returns number of values stored.  STRING values are strdup'ed. */
int object_get(const OBJECT *obj, const char *key, void *ptr, int type, int n, const char *dflt, ...);
int object_getv(const OBJECT *obj, const char *key, void **ptr, int type, int flag);

/* parse the leading "name CLASS { ... }" block of a data file (pio FILEHEADER);
 * returns a malloc'ed OBJECT (not inserted in the database) and the byte offset
 * of the first record in *data_offset. */
OBJECT *object_parse_header(const char *text, long *data_offset);
void object_free(OBJECT *obj);

#ifdef __cplusplus
}
#endif
#endif
