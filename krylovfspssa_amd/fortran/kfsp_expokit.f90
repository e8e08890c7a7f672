! F77-callable names of the dense exponential the reference ships
! (src/expokit/dgpadm.f:2 DGPADM, :171 DGPADMnorm) and its timer
! (src/expokit/clock.f:11 clock), implemented on the library's host routine
! kfsp_padm so that drivers which call them directly keep linking.
SUBROUTINE DGPADM(IDEG, M, T, H, LDH, WSP, LWSP, IPIV, IEXPH, NS, IFLAG)
  USE KFSP_C
  IMPLICIT NONE
  INTEGER :: IDEG, M, LDH, LWSP, IEXPH, NS, IFLAG, IPIV(M)
  DOUBLE PRECISION :: T, H(LDH, M), WSP(LWSP)
  DOUBLE PRECISION :: HNORM
  INTEGER(C_INT) :: RC, NSC
  IFLAG = 0
  IF (LDH < M) IFLAG = -1
  IF (LWSP < 4 * M * M + IDEG + 1) IFLAG = -2
  IF (IFLAG /= 0) STOP 'bad sizes (in input of DGPADM)'
  RC = KFSP_PADM(INT(IDEG, C_INT), INT(M, C_INT), T, H, INT(LDH, C_INT), WSP, NSC, HNORM)
  IF (RC == -3) STOP 'Error - null H in input of DGPADM.'
  IF (RC /= 0) STOP 'Problem in DGESV (within DGPADM)'
  NS = NSC
  IEXPH = 1
END SUBROUTINE DGPADM

SUBROUTINE DGPADMNORM(IDEG, M, T, H, LDH, WSP, LWSP, IPIV, IEXPH, NS, IFLAG, HNORM)
  USE KFSP_C
  IMPLICIT NONE
  INTEGER :: IDEG, M, LDH, LWSP, IEXPH, NS, IFLAG, IPIV(M)
  DOUBLE PRECISION :: T, H(LDH, M), WSP(LWSP), HNORM
  INTEGER(C_INT) :: RC, NSC
  IFLAG = 0
  IF (LDH < M) IFLAG = -1
  IF (LWSP < 4 * M * M + IDEG + 1) IFLAG = -2
  IF (IFLAG /= 0) STOP 'bad sizes (in input of DGPADM)'
  RC = KFSP_PADM(INT(IDEG, C_INT), INT(M, C_INT), T, H, INT(LDH, C_INT), WSP, NSC, HNORM)
  IF (RC == -3) STOP 'Error - null H in input of DGPADM.'
  IF (RC /= 0) STOP 'Problem in DGESV (within DGPADM)'
  NS = NSC
  IEXPH = 1
END SUBROUTINE DGPADMNORM

! wall-clock seconds (the reference's clock.f offers etime and, for threaded
! runs, omp_get_wtime; the host threads of STATESPACE make CPU time meaningless)
DOUBLE PRECISION FUNCTION CLOCK()
  IMPLICIT NONE
  INTEGER(8) :: C, R
  CALL SYSTEM_CLOCK(C, R)
  CLOCK = DBLE(C) / DBLE(R)
END FUNCTION CLOCK
