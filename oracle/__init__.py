"""CPU oracle for the PhaMers count + score path (test infrastructure only)."""
