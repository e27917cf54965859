"""ORACLE - test infrastructure only (see oracle/vit_oracle.py). Never imported by the product."""
