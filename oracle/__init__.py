"""CPU oracle of the TubeDETR hot path - TEST INFRASTRUCTURE ONLY (never imported by tubedetr_amd/)."""
