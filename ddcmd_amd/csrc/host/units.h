/*
 * units.h -- unit handling in the style of simutil's units.c/codata.h (both
 * missing from the reference checkout: src/units.c, src/codata.h are dangling
 * symlinks).  Reconstructed from the call sites:
 *   units_internal(a0, Ry*1e-30/a0^2, 1e-15, e/1e-15, Ry_eV/kB_eV, 1, 1)   ddcMD.c:71
 *   units_external(1e-10, u, 1e-15, e/1e-15, 1, 1, 1)                       ddcMD.c:72
 *   units_convert(value, from, to)  with NULL = internal units, and the
 *   dimension names "l","m","t","i","T" meaning the *external* unit of that
 *   dimension (collection_read.c:94, bioMMFF.c:11,25).
 * Constants are CODATA 2014 (the reference's codata.h values are unknown; the
 * same table is used by the oracle inputs and the device path, so parity does
 * not depend on the choice).
 */
#ifndef DDCMI_UNITS_H
#define DDCMI_UNITS_H
#ifdef __cplusplus
extern "C" {
#endif

/* CODATA 2014, SI */
#define a0_MKS      0.52917721067e-10
#define Rinfhc_MKS  2.179872325e-18
#define Rinfhc_eV   13.605693009
#define kB_MKS      1.38064852e-23
#define kB_eV       8.6173303e-5
#define e_MKS       1.6021766208e-19
#define u_MKS       1.660539040e-27
#define mp_MKS      1.672621898e-27
#define me_MKS      9.10938356e-31
#define NA_MKS      6.022140857e23
#define eps0_MKS    8.854187817e-12

void units_internal(double length, double mass, double time, double current, double temperature, double amount, double luminous);
void units_external(double length, double mass, double time, double current, double temperature, double amount, double luminous);
void units_ddcmd_defaults(void);     /* the two calls of ddcMD.c:71-72 */

/* value * [from] expressed in [to]; NULL means internal units.  Returns NaN
 * and sets units_error() on an unknown symbol or dimension mismatch. */
double units_convert(double value, const char *from, const char *to);
const char *units_error(void);
void units_clear_error(void);      /* units_error() is "" again (a loader clears it, reads its deck, and refuses the deck if it is set) */

/* physical constants in internal units (codata.h: ke, kB) */
double units_ke(void);   /* e^2/(4 pi eps0)  [energy*length/charge^2] */
double units_kB(void);   /* = 1 with the ddcMD internal temperature unit */

#ifdef __cplusplus
}
#endif
#endif
