/* Problem header for 'Synth10Hx' emitted by tools/gen_problem.py. Do not edit.
 * Layout contract: reference iLQG_problem.tem:16-89. */
#ifndef ILQG_PROBLEM_H
#define ILQG_PROBLEM_H

#include <math.h>
#include "mex.h"
#ifndef  HAVE_OCTAVE
#include "matrix.h"
#endif

#define isNANorINF(v) (mxIsNaN(v) || mxIsInf(v))
#define INF mxGetInf()

#define N_X 10
#define N_U 3

#define sizeofQxx 55
#define sizeofQuu 6
#define sizeofQxu 30

/* additive hints for the batched backend (absent in Maxima-generated headers,
 * which are then treated as the general case) */
#define ILQG_PROBLEM_NAME "Synth10Hx"
#define ILQG_STATE_DEPENDENT_LIMITS 1
#define ILQG_TENSOR_NBASIS 20  /* > 0: iLQG_func.c has the factored tensor tables */
#define ILQG_TENSOR_INIT_WRITES 0  /* init_running() writes constant entries of fxx / fuu / fxu */
/* the derivative entries bp_derivsL() writes, X(member, index) each: all others are written once, by init_running() */
#define ILQG_TIME_VARYING(X) X(fx, 0) X(fx, 1) X(fx, 2) X(fx, 3) X(fx, 4) X(fx, 5) X(fx, 6) X(fx, 7) X(fx, 8) X(fx, 9) X(fx, 10) X(fx, 11) X(fx, 12) X(fx, 13) X(fx, 14) X(fx, 15) X(fx, 16) X(fx, 17) X(fx, 18) X(fx, 19) X(fx, 20) X(fx, 21) X(fx, 22) X(fx, 23) X(fx, 24) X(fx, 25) X(fx, 26) X(fx, 27) X(fx, 28) X(fx, 29) X(fx, 30) X(fx, 31) X(fx, 32) X(fx, 33) X(fx, 34) X(fx, 35) X(fx, 36) X(fx, 37) X(fx, 38) X(fx, 39) X(fx, 40) X(fx, 41) X(fx, 42) X(fx, 43) X(fx, 44) X(fx, 45) X(fx, 46) X(fx, 47) X(fx, 48) X(fx, 49) X(fx, 50) X(fx, 51) X(fx, 52) X(fx, 53) X(fx, 54) X(fx, 55) X(fx, 56) X(fx, 57) X(fx, 58) X(fx, 59) X(fx, 60) X(fx, 61) X(fx, 62) X(fx, 63) X(fx, 64) X(fx, 65) X(fx, 66) X(fx, 67) X(fx, 68) X(fx, 69) X(fx, 70) X(fx, 71) X(fx, 72) X(fx, 73) X(fx, 74) X(fx, 75) X(fx, 76) X(fx, 77) X(fx, 78) X(fx, 79) X(fx, 80) X(fx, 81) X(fx, 82) X(fx, 83) X(fx, 84) X(fx, 85) X(fx, 86) X(fx, 87) X(fx, 88) X(fx, 89) X(fx, 90) X(fx, 91) X(fx, 92) X(fx, 93) X(fx, 94) X(fx, 95) X(fx, 96) X(fx, 97) X(fx, 98) X(fx, 99) X(fu, 0) X(fu, 1) X(fu, 2) X(fu, 3) X(fu, 4) X(fu, 5) X(fu, 6) X(fu, 7) X(fu, 8) X(fu, 9) X(fu, 10) X(fu, 11) X(fu, 12) X(fu, 13) X(fu, 14) X(fu, 15) X(fu, 16) X(fu, 17) X(fu, 18) X(fu, 19) X(fu, 20) X(fu, 21) X(fu, 22) X(fu, 23) X(fu, 24) X(fu, 25) X(fu, 26) X(fu, 27) X(fu, 28) X(fu, 29) X(cx, 0) X(cx, 1) X(cx, 2) X(cx, 3) X(cx, 4) X(cx, 5) X(cx, 6) X(cx, 7) X(cx, 8) X(cx, 9) X(cxx, 0) X(cxx, 2) X(cxx, 5) X(cxx, 9) X(cxx, 14) X(cxx, 20) X(cxx, 27) X(cxx, 35) X(cxx, 44) X(cxx, 54) X(cu, 0) X(cu, 1) X(cu, 2)
#if FULL_DDP
#define ILQG_TIME_VARYING_FULL(X) X(fxx, 0) X(fxx, 1) X(fxx, 2) X(fxx, 3) X(fxx, 4) X(fxx, 5) X(fxx, 6) X(fxx, 7) X(fxx, 8) X(fxx, 9) X(fxx, 10) X(fxx, 11) X(fxx, 12) X(fxx, 13) X(fxx, 14) X(fxx, 15) X(fxx, 16) X(fxx, 17) X(fxx, 18) X(fxx, 19) X(fxx, 20) X(fxx, 21) X(fxx, 22) X(fxx, 23) X(fxx, 24) X(fxx, 25) X(fxx, 26) X(fxx, 27) X(fxx, 28) X(fxx, 29) X(fxx, 30) X(fxx, 31) X(fxx, 32) X(fxx, 33) X(fxx, 34) X(fxx, 35) X(fxx, 36) X(fxx, 37) X(fxx, 38) X(fxx, 39) X(fxx, 40) X(fxx, 41) X(fxx, 42) X(fxx, 43) X(fxx, 44) X(fxx, 45) X(fxx, 46) X(fxx, 47) X(fxx, 48) X(fxx, 49) X(fxx, 50) X(fxx, 51) X(fxx, 52) X(fxx, 53) X(fxx, 54) X(fxx, 55) X(fxx, 56) X(fxx, 57) X(fxx, 58) X(fxx, 59) X(fxx, 60) X(fxx, 61) X(fxx, 62) X(fxx, 63) X(fxx, 64) X(fxx, 65) X(fxx, 66) X(fxx, 67) X(fxx, 68) X(fxx, 69) X(fxx, 70) X(fxx, 71) X(fxx, 72) X(fxx, 73) X(fxx, 74) X(fxx, 75) X(fxx, 76) X(fxx, 77) X(fxx, 78) X(fxx, 79) X(fxx, 80) X(fxx, 81) X(fxx, 82) X(fxx, 83) X(fxx, 84) X(fxx, 85) X(fxx, 86) X(fxx, 87) X(fxx, 88) X(fxx, 89) X(fxx, 90) X(fxx, 91) X(fxx, 92) X(fxx, 93) X(fxx, 94) X(fxx, 95) X(fxx, 96) X(fxx, 97) X(fxx, 98) X(fxx, 99) X(fxx, 100) X(fxx, 101) X(fxx, 102) X(fxx, 103) X(fxx, 104) X(fxx, 105) X(fxx, 106) X(fxx, 107) X(fxx, 108) X(fxx, 109) X(fxx, 110) X(fxx, 111) X(fxx, 112) X(fxx, 113) X(fxx, 114) X(fxx, 115) X(fxx, 116) X(fxx, 117) X(fxx, 118) X(fxx, 119) X(fxx, 120) X(fxx, 121) X(fxx, 122) X(fxx, 123) X(fxx, 124) X(fxx, 125) X(fxx, 126) X(fxx, 127) X(fxx, 128) X(fxx, 129) X(fxx, 130) X(fxx, 131) X(fxx, 132) X(fxx, 133) X(fxx, 134) X(fxx, 135) X(fxx, 136) X(fxx, 137) X(fxx, 138) X(fxx, 139) X(fxx, 140) X(fxx, 141) X(fxx, 142) X(fxx, 143) X(fxx, 144) X(fxx, 145) X(fxx, 146) X(fxx, 147) X(fxx, 148) X(fxx, 149) X(fxx, 150) X(fxx, 151) X(fxx, 152) X(fxx, 153) X(fxx, 154) X(fxx, 155) X(fxx, 156) X(fxx, 157) X(fxx, 158) X(fxx, 159) X(fxx, 160) X(fxx, 161) X(fxx, 162) X(fxx, 163) X(fxx, 164) X(fxx, 165) X(fxx, 166) X(fxx, 167) X(fxx, 168) X(fxx, 169) X(fxx, 170) X(fxx, 171) X(fxx, 172) X(fxx, 173) X(fxx, 174) X(fxx, 175) X(fxx, 176) X(fxx, 177) X(fxx, 178) X(fxx, 179) X(fxx, 180) X(fxx, 181) X(fxx, 182) X(fxx, 183) X(fxx, 184) X(fxx, 185) X(fxx, 186) X(fxx, 187) X(fxx, 188) X(fxx, 189) X(fxx, 190) X(fxx, 191) X(fxx, 192) X(fxx, 193) X(fxx, 194) X(fxx, 195) X(fxx, 196) X(fxx, 197) X(fxx, 198) X(fxx, 199) X(fxx, 200) X(fxx, 201) X(fxx, 202) X(fxx, 203) X(fxx, 204) X(fxx, 205) X(fxx, 206) X(fxx, 207) X(fxx, 208) X(fxx, 209) X(fxx, 210) X(fxx, 211) X(fxx, 212) X(fxx, 213) X(fxx, 214) X(fxx, 215) X(fxx, 216) X(fxx, 217) X(fxx, 218) X(fxx, 219) X(fxx, 220) X(fxx, 221) X(fxx, 222) X(fxx, 223) X(fxx, 224) X(fxx, 225) X(fxx, 226) X(fxx, 227) X(fxx, 228) X(fxx, 229) X(fxx, 230) X(fxx, 231) X(fxx, 232) X(fxx, 233) X(fxx, 234) X(fxx, 235) X(fxx, 236) X(fxx, 237) X(fxx, 238) X(fxx, 239) X(fxx, 240) X(fxx, 241) X(fxx, 242) X(fxx, 243) X(fxx, 244) X(fxx, 245) X(fxx, 246) X(fxx, 247) X(fxx, 248) X(fxx, 249) X(fxx, 250) X(fxx, 251) X(fxx, 252) X(fxx, 253) X(fxx, 254) X(fxx, 255) X(fxx, 256) X(fxx, 257) X(fxx, 258) X(fxx, 259) X(fxx, 260) X(fxx, 261) X(fxx, 262) X(fxx, 263) X(fxx, 264) X(fxx, 265) X(fxx, 266) X(fxx, 267) X(fxx, 268) X(fxx, 269) X(fxx, 270) X(fxx, 271) X(fxx, 272) X(fxx, 273) X(fxx, 274) X(fxx, 275) X(fxx, 276) X(fxx, 277) X(fxx, 278) X(fxx, 279) X(fxx, 280) X(fxx, 281) X(fxx, 282) X(fxx, 283) X(fxx, 284) X(fxx, 285) X(fxx, 286) X(fxx, 287) X(fxx, 288) X(fxx, 289) X(fxx, 290) X(fxx, 291) X(fxx, 292) X(fxx, 293) X(fxx, 294) X(fxx, 295) X(fxx, 296) X(fxx, 297) X(fxx, 298) X(fxx, 299) X(fxx, 300) X(fxx, 301) X(fxx, 302) X(fxx, 303) X(fxx, 304) X(fxx, 305) X(fxx, 306) X(fxx, 307) X(fxx, 308) X(fxx, 309) X(fxx, 310) X(fxx, 311) X(fxx, 312) X(fxx, 313) X(fxx, 314) X(fxx, 315) X(fxx, 316) X(fxx, 317) X(fxx, 318) X(fxx, 319) X(fxx, 320) X(fxx, 321) X(fxx, 322) X(fxx, 323) X(fxx, 324) X(fxx, 325) X(fxx, 326) X(fxx, 327) X(fxx, 328) X(fxx, 329) X(fxx, 330) X(fxx, 331) X(fxx, 332) X(fxx, 333) X(fxx, 334) X(fxx, 335) X(fxx, 336) X(fxx, 337) X(fxx, 338) X(fxx, 339) X(fxx, 340) X(fxx, 341) X(fxx, 342) X(fxx, 343) X(fxx, 344) X(fxx, 345) X(fxx, 346) X(fxx, 347) X(fxx, 348) X(fxx, 349) X(fxx, 350) X(fxx, 351) X(fxx, 352) X(fxx, 353) X(fxx, 354) X(fxx, 355) X(fxx, 356) X(fxx, 357) X(fxx, 358) X(fxx, 359) X(fxx, 360) X(fxx, 361) X(fxx, 362) X(fxx, 363) X(fxx, 364) X(fxx, 365) X(fxx, 366) X(fxx, 367) X(fxx, 368) X(fxx, 369) X(fxx, 370) X(fxx, 371) X(fxx, 372) X(fxx, 373) X(fxx, 374) X(fxx, 375) X(fxx, 376) X(fxx, 377) X(fxx, 378) X(fxx, 379) X(fxx, 380) X(fxx, 381) X(fxx, 382) X(fxx, 383) X(fxx, 384) X(fxx, 385) X(fxx, 386) X(fxx, 387) X(fxx, 388) X(fxx, 389) X(fxx, 390) X(fxx, 391) X(fxx, 392) X(fxx, 393) X(fxx, 394) X(fxx, 395) X(fxx, 396) X(fxx, 397) X(fxx, 398) X(fxx, 399) X(fxx, 400) X(fxx, 401) X(fxx, 402) X(fxx, 403) X(fxx, 404) X(fxx, 405) X(fxx, 406) X(fxx, 407) X(fxx, 408) X(fxx, 409) X(fxx, 410) X(fxx, 411) X(fxx, 412) X(fxx, 413) X(fxx, 414) X(fxx, 415) X(fxx, 416) X(fxx, 417) X(fxx, 418) X(fxx, 419) X(fxx, 420) X(fxx, 421) X(fxx, 422) X(fxx, 423) X(fxx, 424) X(fxx, 425) X(fxx, 426) X(fxx, 427) X(fxx, 428) X(fxx, 429) X(fxx, 430) X(fxx, 431) X(fxx, 432) X(fxx, 433) X(fxx, 434) X(fxx, 435) X(fxx, 436) X(fxx, 437) X(fxx, 438) X(fxx, 439) X(fxx, 440) X(fxx, 441) X(fxx, 442) X(fxx, 443) X(fxx, 444) X(fxx, 445) X(fxx, 446) X(fxx, 447) X(fxx, 448) X(fxx, 449) X(fxx, 450) X(fxx, 451) X(fxx, 452) X(fxx, 453) X(fxx, 454) X(fxx, 455) X(fxx, 456) X(fxx, 457) X(fxx, 458) X(fxx, 459) X(fxx, 460) X(fxx, 461) X(fxx, 462) X(fxx, 463) X(fxx, 464) X(fxx, 465) X(fxx, 466) X(fxx, 467) X(fxx, 468) X(fxx, 469) X(fxx, 470) X(fxx, 471) X(fxx, 472) X(fxx, 473) X(fxx, 474) X(fxx, 475) X(fxx, 476) X(fxx, 477) X(fxx, 478) X(fxx, 479) X(fxx, 480) X(fxx, 481) X(fxx, 482) X(fxx, 483) X(fxx, 484) X(fxx, 485) X(fxx, 486) X(fxx, 487) X(fxx, 488) X(fxx, 489) X(fxx, 490) X(fxx, 491) X(fxx, 492) X(fxx, 493) X(fxx, 494) X(fxx, 495) X(fxx, 496) X(fxx, 497) X(fxx, 498) X(fxx, 499) X(fxx, 500) X(fxx, 501) X(fxx, 502) X(fxx, 503) X(fxx, 504) X(fxx, 505) X(fxx, 506) X(fxx, 507) X(fxx, 508) X(fxx, 509) X(fxx, 510) X(fxx, 511) X(fxx, 512) X(fxx, 513) X(fxx, 514) X(fxx, 515) X(fxx, 516) X(fxx, 517) X(fxx, 518) X(fxx, 519) X(fxx, 520) X(fxx, 521) X(fxx, 522) X(fxx, 523) X(fxx, 524) X(fxx, 525) X(fxx, 526) X(fxx, 527) X(fxx, 528) X(fxx, 529) X(fxx, 530) X(fxx, 531) X(fxx, 532) X(fxx, 533) X(fxx, 534) X(fxx, 535) X(fxx, 536) X(fxx, 537) X(fxx, 538) X(fxx, 539) X(fxx, 540) X(fxx, 541) X(fxx, 542) X(fxx, 543) X(fxx, 544) X(fxx, 545) X(fxx, 546) X(fxx, 547) X(fxx, 548) X(fxx, 549) X(fuu, 0) X(fuu, 1) X(fuu, 2) X(fuu, 3) X(fuu, 4) X(fuu, 5) X(fuu, 6) X(fuu, 7) X(fuu, 8) X(fuu, 9) X(fuu, 10) X(fuu, 11) X(fuu, 12) X(fuu, 13) X(fuu, 14) X(fuu, 15) X(fuu, 16) X(fuu, 17) X(fuu, 18) X(fuu, 19) X(fuu, 20) X(fuu, 21) X(fuu, 22) X(fuu, 23) X(fuu, 24) X(fuu, 25) X(fuu, 26) X(fuu, 27) X(fuu, 28) X(fuu, 29) X(fuu, 30) X(fuu, 31) X(fuu, 32) X(fuu, 33) X(fuu, 34) X(fuu, 35) X(fuu, 36) X(fuu, 37) X(fuu, 38) X(fuu, 39) X(fuu, 40) X(fuu, 41) X(fuu, 42) X(fuu, 43) X(fuu, 44) X(fuu, 45) X(fuu, 46) X(fuu, 47) X(fuu, 48) X(fuu, 49) X(fuu, 50) X(fuu, 51) X(fuu, 52) X(fuu, 53) X(fuu, 54) X(fuu, 55) X(fuu, 56) X(fuu, 57) X(fuu, 58) X(fuu, 59) X(fxu, 0) X(fxu, 1) X(fxu, 2) X(fxu, 3) X(fxu, 4) X(fxu, 5) X(fxu, 6) X(fxu, 7) X(fxu, 8) X(fxu, 9) X(fxu, 10) X(fxu, 11) X(fxu, 12) X(fxu, 13) X(fxu, 14) X(fxu, 15) X(fxu, 16) X(fxu, 17) X(fxu, 18) X(fxu, 19) X(fxu, 20) X(fxu, 21) X(fxu, 22) X(fxu, 23) X(fxu, 24) X(fxu, 25) X(fxu, 26) X(fxu, 27) X(fxu, 28) X(fxu, 29) X(fxu, 30) X(fxu, 31) X(fxu, 32) X(fxu, 33) X(fxu, 34) X(fxu, 35) X(fxu, 36) X(fxu, 37) X(fxu, 38) X(fxu, 39) X(fxu, 40) X(fxu, 41) X(fxu, 42) X(fxu, 43) X(fxu, 44) X(fxu, 45) X(fxu, 46) X(fxu, 47) X(fxu, 48) X(fxu, 49) X(fxu, 50) X(fxu, 51) X(fxu, 52) X(fxu, 53) X(fxu, 54) X(fxu, 55) X(fxu, 56) X(fxu, 57) X(fxu, 58) X(fxu, 59) X(fxu, 60) X(fxu, 61) X(fxu, 62) X(fxu, 63) X(fxu, 64) X(fxu, 65) X(fxu, 66) X(fxu, 67) X(fxu, 68) X(fxu, 69) X(fxu, 70) X(fxu, 71) X(fxu, 72) X(fxu, 73) X(fxu, 74) X(fxu, 75) X(fxu, 76) X(fxu, 77) X(fxu, 78) X(fxu, 79) X(fxu, 80) X(fxu, 81) X(fxu, 82) X(fxu, 83) X(fxu, 84) X(fxu, 85) X(fxu, 86) X(fxu, 87) X(fxu, 88) X(fxu, 89) X(fxu, 90) X(fxu, 91) X(fxu, 92) X(fxu, 93) X(fxu, 94) X(fxu, 95) X(fxu, 96) X(fxu, 97) X(fxu, 98) X(fxu, 99) X(fxu, 100) X(fxu, 101) X(fxu, 102) X(fxu, 103) X(fxu, 104) X(fxu, 105) X(fxu, 106) X(fxu, 107) X(fxu, 108) X(fxu, 109) X(fxu, 110) X(fxu, 111) X(fxu, 112) X(fxu, 113) X(fxu, 114) X(fxu, 115) X(fxu, 116) X(fxu, 117) X(fxu, 118) X(fxu, 119) X(fxu, 120) X(fxu, 121) X(fxu, 122) X(fxu, 123) X(fxu, 124) X(fxu, 125) X(fxu, 126) X(fxu, 127) X(fxu, 128) X(fxu, 129) X(fxu, 130) X(fxu, 131) X(fxu, 132) X(fxu, 133) X(fxu, 134) X(fxu, 135) X(fxu, 136) X(fxu, 137) X(fxu, 138) X(fxu, 139) X(fxu, 140) X(fxu, 141) X(fxu, 142) X(fxu, 143) X(fxu, 144) X(fxu, 145) X(fxu, 146) X(fxu, 147) X(fxu, 148) X(fxu, 149) X(fxu, 150) X(fxu, 151) X(fxu, 152) X(fxu, 153) X(fxu, 154) X(fxu, 155) X(fxu, 156) X(fxu, 157) X(fxu, 158) X(fxu, 159) X(fxu, 160) X(fxu, 161) X(fxu, 162) X(fxu, 163) X(fxu, 164) X(fxu, 165) X(fxu, 166) X(fxu, 167) X(fxu, 168) X(fxu, 169) X(fxu, 170) X(fxu, 171) X(fxu, 172) X(fxu, 173) X(fxu, 174) X(fxu, 175) X(fxu, 176) X(fxu, 177) X(fxu, 178) X(fxu, 179) X(fxu, 180) X(fxu, 181) X(fxu, 182) X(fxu, 183) X(fxu, 184) X(fxu, 185) X(fxu, 186) X(fxu, 187) X(fxu, 188) X(fxu, 189) X(fxu, 190) X(fxu, 191) X(fxu, 192) X(fxu, 193) X(fxu, 194) X(fxu, 195) X(fxu, 196) X(fxu, 197) X(fxu, 198) X(fxu, 199) X(fxu, 200) X(fxu, 201) X(fxu, 202) X(fxu, 203) X(fxu, 204) X(fxu, 205) X(fxu, 206) X(fxu, 207) X(fxu, 208) X(fxu, 209) X(fxu, 210) X(fxu, 211) X(fxu, 212) X(fxu, 213) X(fxu, 214) X(fxu, 215) X(fxu, 216) X(fxu, 217) X(fxu, 218) X(fxu, 219) X(fxu, 220) X(fxu, 221) X(fxu, 222) X(fxu, 223) X(fxu, 224) X(fxu, 225) X(fxu, 226) X(fxu, 227) X(fxu, 228) X(fxu, 229) X(fxu, 230) X(fxu, 231) X(fxu, 232) X(fxu, 233) X(fxu, 234) X(fxu, 235) X(fxu, 236) X(fxu, 237) X(fxu, 238) X(fxu, 239) X(fxu, 240) X(fxu, 241) X(fxu, 242) X(fxu, 243) X(fxu, 244) X(fxu, 245) X(fxu, 246) X(fxu, 247) X(fxu, 248) X(fxu, 249) X(fxu, 250) X(fxu, 251) X(fxu, 252) X(fxu, 253) X(fxu, 254) X(fxu, 255) X(fxu, 256) X(fxu, 257) X(fxu, 258) X(fxu, 259) X(fxu, 260) X(fxu, 261) X(fxu, 262) X(fxu, 263) X(fxu, 264) X(fxu, 265) X(fxu, 266) X(fxu, 267) X(fxu, 268) X(fxu, 269) X(fxu, 270) X(fxu, 271) X(fxu, 272) X(fxu, 273) X(fxu, 274) X(fxu, 275) X(fxu, 276) X(fxu, 277) X(fxu, 278) X(fxu, 279) X(fxu, 280) X(fxu, 281) X(fxu, 282) X(fxu, 283) X(fxu, 284) X(fxu, 285) X(fxu, 286) X(fxu, 287) X(fxu, 288) X(fxu, 289) X(fxu, 290) X(fxu, 291) X(fxu, 292) X(fxu, 293) X(fxu, 294) X(fxu, 295) X(fxu, 296) X(fxu, 297) X(fxu, 298) X(fxu, 299)
#else
#define ILQG_TIME_VARYING_FULL(X)
#endif
/* among the others: the entries that are identically 0 (what a dense back_pass multiplies by zero, matMult.c:3-72) */
#define ILQG_STRUCTURAL_ZERO(X) X(cxx, 1) X(cxx, 3) X(cxx, 4) X(cxx, 6) X(cxx, 7) X(cxx, 8) X(cxx, 10) X(cxx, 11) X(cxx, 12) X(cxx, 13) X(cxx, 15) X(cxx, 16) X(cxx, 17) X(cxx, 18) X(cxx, 19) X(cxx, 21) X(cxx, 22) X(cxx, 23) X(cxx, 24) X(cxx, 25) X(cxx, 26) X(cxx, 28) X(cxx, 29) X(cxx, 30) X(cxx, 31) X(cxx, 32) X(cxx, 33) X(cxx, 34) X(cxx, 36) X(cxx, 37) X(cxx, 38) X(cxx, 39) X(cxx, 40) X(cxx, 41) X(cxx, 42) X(cxx, 43) X(cxx, 45) X(cxx, 46) X(cxx, 47) X(cxx, 48) X(cxx, 49) X(cxx, 50) X(cxx, 51) X(cxx, 52) X(cxx, 53) X(cuu, 1) X(cuu, 3) X(cuu, 4) X(cxu, 0) X(cxu, 1) X(cxu, 2) X(cxu, 3) X(cxu, 4) X(cxu, 5) X(cxu, 6) X(cxu, 7) X(cxu, 8) X(cxu, 9) X(cxu, 10) X(cxu, 11) X(cxu, 12) X(cxu, 13) X(cxu, 14) X(cxu, 15) X(cxu, 16) X(cxu, 17) X(cxu, 18) X(cxu, 19) X(cxu, 20) X(cxu, 21) X(cxu, 22) X(cxu, 23) X(cxu, 24) X(cxu, 25) X(cxu, 26) X(cxu, 27) X(cxu, 28) X(cxu, 29)
#if FULL_DDP
#define ILQG_STRUCTURAL_ZERO_FULL(X) 
#else
#define ILQG_STRUCTURAL_ZERO_FULL(X)
#endif

typedef struct {
    double x[N_X];
    double u[N_U];
    double lower[N_U];
    double upper[N_U];
    double lower_sign[N_U];
    double upper_sign[N_U];
    double lower_hx[N_X*N_U];
    double upper_hx[N_X*N_U];

    double l[N_U];
    double L[N_U*N_X];
    double c;
    double cx[N_X];
    double cxx[sizeofQxx];
    double cu[N_U];
    double cuu[sizeofQuu];
    double cxu[sizeofQxu];
    double fx[N_X*N_X];
    double fu[N_X*N_U];
#if FULL_DDP
    double fxx[N_X*sizeofQxx];
    double fuu[N_X*sizeofQuu];
    double fxu[N_X*sizeofQxu];
#endif
    double s1_0;
    double s1_1;
    double s1_2;
    double s1_3;
    double s1_4;
    double s1_5;
    double s1_6;
    double s1_7;
    double s1_8;
    double s1_9;
    double s2_0;
    double s2_1;
    double s2_2;
    double s2_3;
    double s2_4;
    double s2_5;
    double s2_6;
    double s2_7;
    double s2_8;
    double s2_9;
#if FULL_DDP
#endif
} trajEl_t;

typedef struct {
    double x[N_X];

    double c;
    double cx[N_X];
    double cxx[sizeofQxx];
} trajFin_t;

typedef struct {
    trajEl_t* t;
    trajFin_t f;
} traj_t;

typedef struct {
} multipliersEl_t;

typedef struct {
} multipliersFin_t;

typedef struct {
    multipliersEl_t* t;
    multipliersFin_t f;
} multipliers_t;

#endif // ILQG_PROBLEM_H
